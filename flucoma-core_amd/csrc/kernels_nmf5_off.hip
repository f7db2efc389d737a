// kernels_nmf5_off.hip -- the off-size instantiations of nmf_update5_kernel (compute ranks 24 | 40, 48, 56 | 72 .. 112 in eights on the
// arrays of rank 32 / 64 / 128) as their own translation unit: the kernel template and its launchers are kernels_nmf5.hip's,
// the instantiation list is split in two so that the halves compile side by side (flucoma-core_amd/build.py).
#define FLUHIP_K5_OFFSIZE_TU 1
#include "kernels_nmf5.hip"
