"""TEST INFRASTRUCTURE ONLY -- the CPU oracle of the SE-SSD inference hot path.

Nothing under `oracle/` may be imported by the product (`se-ssd_amd/`). Allowed
importers: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg.

Parity status per piece (see DESIGN.md "Oracle"):
  voxelize.c / vfe            pinned to the reference's own Python source (golden vectors)
  iou3d.c                     pinned to the COMPILED reference (oracle/_ref) + golden vectors
  rotate_nms.c                helpers pinned to reference numpy source; polygon IoU cross-checked
                              against oracle/_ref (boost::geometry is absent: nms_cpu.h cannot be built)
  sparse_conv.py              PARITY UNPINNED -- spconv v1 is a third-party dependency that is
                              not vendored in /root/reference; semantics restated and cross-checked
                              against torch.nn.functional.conv3d on the densified grid
  dense_head.py, postprocess.py  pinned to the reference's own Python source (golden vectors)
"""
from .capi import *  # noqa: F401,F403
