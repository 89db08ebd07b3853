// met2_fit_nnls_lcurve.hip -- explicit instantiations of the fit kernel for one family of methods (fit_kernel.hpp); empty unless -DMET2_SPLIT_TU.
#ifdef MET2_SPLIT_TU
#include "fit_kernel.hpp"
template int launch_fit_nb<0, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<0, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<1, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<1, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<3, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<3, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<0, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<0, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<1, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<1, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<3, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<3, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
#endif
