"""MI355X-native per-voxel regularised-NNLS T2-spectrum solver (drop-in for the hot path of
ejcanalesr/multicomponent-T2-toolbox).  Package layout:

  csrc/                     HIP kernels + the C ABI of include/met2_hip.h
  plan.py                   Met2Plan (ctypes face of the C ABI; torch for device memory only)
  intravoxel_algorithms.py  nnls, nnls_tik, nnls_x2, nnls_lcurve_wrapper, nnls_gcv, BayesReg_nnls
  epg.py                    create_Dic_3D, create_met2_design_matrix_epg
  flip_angle_algorithms.py  compute_optimal_FA, fitting_slice_FA_brute_force
  motor.py                  create_Laplacian_matrix, fitting_slice_T2, recon_met2_arrays (voxel loop), nesma_filter, gaussian_smooth, ROI mode
  tv.py                     tv_denoise_volume / tv_chambolle: denoise='TV' of the driver through met2_tv_chambolle (csrc/met2_tv.hip)
  nifti.py                  NIfTI-1 reader / writer for the driver's on-disk contract
  dist.py                   one-process-per-GPU voxel sharding + the single gather of output maps
  synth.py                  seeded synthetic volumes (the reference's Monte-Carlo recipe)

There is no CPU fallback: without the built HIP library and a visible GPU every call raises."""
from ._lib import Met2Error  # noqa: F401
from .plan import MAP_NAMES, METHODS, PENALTIES, Met2Plan  # noqa: F401
