import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refonly: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "variants: A/B switch variants of the default path; run only with VDQN_TEST_VARIANTS=1 (tools/job.sh variants)")
    config.addinivalue_line("markers", "slow_oracle: minutes of CPU-oracle time at a full-size geometry; run only with VDQN_TEST_SLOW=1 (tools/job.sh slow)")


# Collection order of the GPU suite: a time limit must cut the multi-process tests, never the tests that
# isolate a kernel.  First one quick test per SURVEY section-8 row (ROW_FIRST), then per-op oracle tests -> golden / engine step
# tests -> full-size -> boundary -> multi-process.  Files not named here keep their place at the front (CPU tests).
FILE_ORDER = ["test_gpu_ops", "test_gpu_skinny", "test_inverse_model", "test_gpu_engine", "test_gpu_basic", "test_gpu_autograd",
              "test_gpu_fullsize", "test_gpu_convergence", "test_gpu_boundary", "test_gpu_ddp", "test_gpu_launch"]
ROW_FIRST = [
    "test_gpu_ops.py::test_td_loss_branches_vs_golden",                         # a2 process_batch
    "test_gpu_engine.py::test_forward_matches_reference_golden",                # a3 forward (G2)
    "test_gpu_ops.py::test_conv_forward[case8-dtype1]",                         # a4 trunk conv, layer1 geometry, bf16
    "test_gpu_skinny.py::test_features8_valid_conv_forward",                    # a5 head
    "test_gpu_engine.py::test_td_steps_match_reference_golden_f32",             # a1/a6/a7/a8 loop body, set_train, backward, Adam (G3)
    "test_gpu_engine.py::test_td_step_matches_oracle_all_elements[f32-0.001-0.001-101]",         # a7 every gradient element
    "test_gpu_ops.py::test_adam_matches_torch",                                 # a8
    "test_gpu_engine.py::test_target_sync_timing",                              # a9
    "test_gpu_engine.py::test_param_table_matches_reference_layout",            # a10
    "test_gpu_boundary.py::test_module_forward_eval_and_b1_quirk",              # f2 / b-outer
    "test_gpu_boundary.py::test_host_frame_stream_device_branch",               # f1 streaming input path
    "test_gpu_basic.py::test_basic_td_steps_match_reference_golden_f32",        # f3
    "test_inverse_model.py::test_gpu_forward_matches_reference_golden",         # f4
    "test_gpu_ddp.py::test_n_ranks_equal_one_big_batch[world2]",                      # e
]


def _rank(item):
    nid = item.nodeid
    for i, key in enumerate(ROW_FIRST):
        if key in nid:
            return (0, i)
    stem = os.path.basename(item.fspath.strpath if hasattr(item.fspath, "strpath") else str(item.fspath))[:-3]
    if stem in FILE_ORDER:
        return (2 + FILE_ORDER.index(stem), 0)
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    for marker, env in (("variants", "VDQN_TEST_VARIANTS"), ("slow_oracle", "VDQN_TEST_SLOW")):
        if os.environ.get(env, "0") != "1":
            keep, drop = [], []
            for it in items:
                (drop if it.get_closest_marker(marker) else keep).append(it)
            if drop:
                config.hook.pytest_deselected(items=drop)
                items[:] = keep
    order = {id(it): n for n, it in enumerate(items)}
    items.sort(key=lambda it: (_rank(it), order[id(it)]))


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def g1():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "g1_state_dict.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_basic():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_basic.npz"), allow_pickle=False)
