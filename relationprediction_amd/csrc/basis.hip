// Basis-decomposition path (BasisGcn, code/encoders/message_gcns/gcn_basis.py) -- placeholder
// translation unit; the kernels land in the next milestone.
#include "rgcn_internal.h"

namespace rgcn {

rgcn_status basis_aggregate_forward(rgcn_ctx* c, int, const float*, float*) {
  RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "basis path not built in this revision");
}
rgcn_status basis_backward_sparse(rgcn_ctx* c, int, const float*, const float*, float*) {
  RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "basis path not built in this revision");
}
rgcn_status basis_to_device_layout(rgcn_ctx* c, const float*, float*) {
  RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "basis path not built in this revision");
}
rgcn_status basis_from_device_layout(rgcn_ctx* c, const float*, float*) {
  RGCN_FAIL(c, RGCN_ERR_UNSUPPORTED, "basis path not built in this revision");
}

}  // namespace rgcn
