// cim_spec.hip — the CIM step kernels specialised for ONE plan (topology + config): every integer dimension and layout
// offset of CimParams is a compile-time constant (cim_spec_dims.h = the text of mrx_cim_plan_defines, generated next to a
// copy of this file by maro_amd/cim/specialize.py, which compiles it to a gfx950 code object with
//   hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I<dir of cim_spec_dims.h> -I maro_amd/csrc cim_spec.hip
// and hands the image to mrx_cim_load_step_kernels).  Same device source as the generic build (cim_device.h); only KD() changes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave.h"
#define MRX_SPECIALIZED 1
#include "cim_spec_dims.h"
#include "cim_prof.h"
#include "cim_device.h"

#ifndef MRX_STEP_WAVES
#define MRX_STEP_WAVES 3  // specialised: ~120 VGPRs; LDS (9 waves/CU) is the occupancy limit, so 3 waves/SIMD is enough
#endif
#include "cim_step_kernels.h"
