// TEST INFRASTRUCTURE ONLY. Translation unit that compiles the REFERENCE'S det3d/ops/nms/nms_cpu.h where it lies (nothing is
// copied) into a Python module, with <boost/geometry.hpp> resolved to oracle/boost_shim (boost is not installed here; see the
// shim's header for what that substitution does and does not pin). The reference's own module (nms.cc) also needs its CUDA
// kernel; this stub binds the three CPU functions of that header.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/numpy.h>
#include <cmath>
#include <cstdlib>
#include <iostream>
namespace py = pybind11;
using namespace pybind11::literals;
using std::exp;
using std::pow;
using std::sqrt;
#include REF_NMS_CPU_H

PYBIND11_MODULE(ref_nms_cpu, m) {
  m.def("non_max_suppression_cpu", &non_max_suppression_cpu<float>, "boxes"_a, "order"_a, "nms_overlap_thresh"_a, "eps"_a);
  m.def("non_max_suppression_cpu", &non_max_suppression_cpu<double>, "boxes"_a, "order"_a, "nms_overlap_thresh"_a, "eps"_a);
  m.def("rotate_non_max_suppression_cpu", &rotate_non_max_suppression_cpu<float>, "box_corners"_a, "order"_a, "standup_iou"_a, "thresh"_a);
  m.def("rotate_non_max_suppression_cpu", &rotate_non_max_suppression_cpu<double>, "box_corners"_a, "order"_a, "standup_iou"_a, "thresh"_a);
  m.def("IOU_weighted_rotate_non_max_suppression_cpu", &IOU_weighted_rotate_non_max_suppression_cpu<float>);
  m.def("IOU_weighted_rotate_non_max_suppression_cpu", &IOU_weighted_rotate_non_max_suppression_cpu<double>);
}
