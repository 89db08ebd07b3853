// met2_fit_x2_nb2.hip -- explicit instantiations of the fit kernel for one family of methods (fit_kernel.hpp); empty unless -DMET2_SPLIT_TU.
#ifdef MET2_SPLIT_TU
#include "fit_kernel.hpp"
template int launch_fit_nb<2, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
template int launch_fit_nb<12, 2, false>(const FitArgs &, const LaunchGeom &, hipStream_t);
#endif
