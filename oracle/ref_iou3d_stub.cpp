// TEST INFRASTRUCTURE ONLY.
// C-ABI wrappers around the REFERENCE's own CPU implementation
// /root/reference/det3d/core/iou3d/src/iou3d_cpu.cpp (compiled from where it lies by
// oracle/build.py into oracle/_ref/libref_iou3d.so; never copied into this repo).
// The reference entry points take at::Tensor; these wrappers view raw float buffers.
#include <torch/torch.h>

int boxes_overlap_bev_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_overlap);
int boxes_iou_bev_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou);
int boxes_iou3d_cpu(at::Tensor boxes_a, at::Tensor boxes_b, at::Tensor ans_iou);

static at::Tensor view(float* p, int64_t r, int64_t c) {
  return torch::from_blob(p, {r, c}, torch::TensorOptions().dtype(torch::kFloat32));
}

extern "C" {
int ref_boxes_overlap_bev_cpu(float* a, int n, float* b, int m, float* out) {
  return boxes_overlap_bev_cpu(view(a, n, 5), view(b, m, 5), view(out, n, m));
}
int ref_boxes_iou_bev_cpu(float* a, int n, float* b, int m, float* out) {
  return boxes_iou_bev_cpu(view(a, n, 5), view(b, m, 5), view(out, n, m));
}
int ref_boxes_iou3d_cpu(float* a, int n, float* b, int m, float* out) {
  return boxes_iou3d_cpu(view(a, n, 7), view(b, m, 7), view(out, n, m));
}
}
