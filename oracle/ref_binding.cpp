// oracle/ref_binding.cpp -- TEST INFRASTRUCTURE. Python binding for the reference's OWN hash-grid operators, built by
// oracle/ref_build.py into oracle/_ref/shacira_ref_ops.so. This file contains no arithmetic: it includes the reference's
// header where it lies (-I/root/reference/wisp/csrc/ops) and exports the four functions under the names the reference's
// wisp/csrc/bindings.cpp:24-28 gives them (that file cannot be built here: it pulls in every other wisp / kaolin op).
#include <torch/extension.h>

#include "hashgrid_interpolate.h"  // the reference's header (reference wisp/csrc/ops/hashgrid_interpolate.h:18-50)

PYBIND11_MODULE(shacira_ref_ops, m) {
    m.def("hashgrid_interpolate_cuda", &wisp::hashgrid_interpolate_cuda);
    m.def("hashgrid_interpolate_backward_cuda", &wisp::hashgrid_interpolate_backward_cuda);
    m.def("hashgrid_interpolate2d_cuda", &wisp::hashgrid_interpolate2d_cuda);
    m.def("hashgrid_interpolate2d_backward_cuda", &wisp::hashgrid_interpolate2d_backward_cuda);
}
