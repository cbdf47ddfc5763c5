// tests/hipemu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY: the emulator's thread coordinates for libcfhd_amd_hipemu.so (hip/hip_runtime.h).
#include "hip_emu.h"
dim3 threadIdx, blockIdx, blockDim, gridDim;
