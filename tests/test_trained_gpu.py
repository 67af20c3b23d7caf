"""A small instance of tests/trained_parity.py (round-4 review items "Missing 1 - 2": nothing trained, no parity on trained weights):
300 captured SE-SSD iterations on fresh synthetic batches, then the trained student through the HIP engine and through the CPU
oracle on held-out scans. The full-size record (2000 iterations, 200 held-out scans) is profiles/r5_trained_parity.json."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_training_lowers_the_loss_and_the_trained_engine_equals_the_oracle(dev):
    import trained_parity
    out, step = trained_parity.run(dev, iterations=300, scenes=48, heldout=12, batch=4, seed=0, log_every=20, workers=0)
    tr, ck = out["training"], out["training_checks"]
    print({k: v for k, v in tr.items() if k != "what"}, ck)
    print(out["ap_table_engine"])
    # the loop ran on fresh batches with the data path inside the clock and never overflowed a capacity
    assert tr["timed_iterations"] == 299 and tr["overflow_flags"] == 0 and tr["sparse_overflow_flag"] == 0 and tr["samples_per_s"] > 50
    # the loss goes down (moving average of the logged iterations) and the late iterations exercise positives
    assert ck["loss_last_window"] < 0.8 * ck["loss_first_window"], ck
    assert ck["moving_average_first_last"][1] < ck["moving_average_first_last"][0]
    assert ck["late_positives_min"] > 0 and ck["late_matched_boxes_mean"] >= 0
    # the teacher is the EMA of the student (trainer_sessd.py:315-318), carried beside the fused update in torch arithmetic
    assert tr["teacher_vs_ema_of_student_maxabs"] <= 1e-5 * (1.0 + tr["teacher_maxabs"]), tr
    assert out["largest_parameter_or_buffer_change"] > 1e-3
    # trained weights: car-sized boxes, so the STRICT comparison rule applies -- every held-out frame
    cmp = out["engine_vs_oracle_strict"]
    assert cmp["ok"] and cmp["frames"] == 12, cmp
    assert max(out["detected_box_sizes_max_m"]) < 20.0
    assert len(out["active_tile_layers_of_the_engine"]) == 10
    # both detection sets through the KITTI evaluation: the same AP
    for k, v in out["ap_abs_difference"].items():
        assert max(v) <= 0.1, (k, v, out["ap_engine"], out["ap_oracle"])
    assert out["ap_engine"]["detections"] > 0
