// Dense f32-MFMA path (placeholder until the kernels land): reports "unsupported"
// so PP_METHOD_AUTO resolves to the gather / sort kernels.
#include "pp_internal.h"

bool pp_dense_supported(const PPShape&) { return false; }
size_t pp_dense_bwd_weight_workspace_bytes(const PPShape&) { return 0; }
hipError_t pp_launch_fwd_dense(const float*, const float*, float*, int32_t*, const PPShape&,
                               const BasisTab&, hipStream_t) { return hipErrorNotSupported; }
hipError_t pp_launch_bwd_input_dense(const float*, const float*, const float*, float*,
                                     const PPShape&, const BasisTab&, hipStream_t) { return hipErrorNotSupported; }
hipError_t pp_launch_bwd_weight_dense(const float*, const float*, float*, void*, const PPShape&,
                                      const BasisTab&, hipStream_t) { return hipErrorNotSupported; }
