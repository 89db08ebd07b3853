// met2_fit_gcv.hip -- explicit instantiations of the fit kernel for one family of methods (fit_kernel.hpp); empty unless -DMET2_SPLIT_TU.
#ifdef MET2_SPLIT_TU
#include "fit_kernel.hpp"
template int launch_fit_nb<4, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<4, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<6, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<6, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<14, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<14, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<16, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<16, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<4, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<4, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<6, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<6, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
#endif
