// met2_fit_bayes.hip -- explicit instantiations of the fit kernel for one family of methods (fit_kernel.hpp); empty unless -DMET2_SPLIT_TU.
#ifdef MET2_SPLIT_TU
#include "fit_kernel.hpp"
template int launch_fit_nb<5, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<5, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<15, 1, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<15, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<5, 1, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<5, 2, true>(const FitArgs &, const LaunchGeom &, hipStream_t);
#endif
