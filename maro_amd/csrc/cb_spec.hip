// cb_spec.hip — the citi_bike reset and step kernels specialised for ONE plan (topology + config + batch size): every integer
// dimension of CbParams is a compile-time constant (cb_spec_dims.h = the text of mrx_cb_plan_defines, generated next to this
// file's compile by maro_amd/cim/specialize.py: hipcc --genco --offload-arch=gfx950 -O3 ..., loaded by mrx_cb_load_step_kernels).
// Same device source as the generic build (cb_device.h); only CD() / CDA() change.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/maro_amd_citi_bike.h"
#include "wave.h"
#define MRX_SPECIALIZED 1
#include "cb_spec_dims.h"
#include "cb_device.h"
#include "cb_wave.h"
#include "cb_step_kernels.h"
