"""CPU-side tests (no GPU): host graph packing vs the oracle, the C-ABI library's exports, and the
loud-failure rule of the product path."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import student_mechanism_design_amd as sy
from oracle import oracle_lib as ol
from tests.helpers import load_trace, trace_index

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sampler_follows_reference_statistics():
    """graph_layout.py:9-52: connected, requested edge count, weights in 1..4, no duplicate edges,
    extras respect the degree cap of 4 (tree edges may exceed it)."""
    rng = np.random.default_rng(0)
    for n, e in ((15, 20), (50, 90), (200, 400)):
        b = sy.sample_board(n, e, rng=rng)
        assert b.num_nodes == n and b.num_edges == e
        assert b.edges.min() >= 1 and b.edges.max() <= 4
        und = {tuple(sorted(x)) for x in b.edge_links.tolist()}
        assert len(und) == e
        d = sy.all_pairs_shortest_paths(b)  # raises if disconnected
        assert d.shape == (n, n)
        tree, extra = b.edge_links[: n - 1], b.edge_links[n - 1:]
        deg = np.bincount(tree.reshape(-1), minlength=n)
        for u, v in extra:
            assert deg[u] < 4 and deg[v] < 4
            deg[u] += 1
            deg[v] += 1
    tree_only = sy.sample_board(12, None, rng=rng)
    assert tree_only.num_edges == 11


def test_pool_has_a_common_edge_count():
    boards = sy.sample_board_pool(5, 40, 70, seed=3)
    assert len({b.num_edges for b in boards}) == 1
    sat = sy.sample_board_pool(3, 30, 70, seed=1)  # unreachable request: first sample fixes the count
    assert len({b.num_edges for b in sat}) == 1 and sat[0].num_edges < 70


def test_ell_and_apsp_match_oracle_on_golden_boards():
    for e in trace_index()[::3]:
        tr = load_trace(e["file"])
        n = int(tr["N"])
        board = sy.make_board(n, tr["edge_links"], tr["edge_w"])
        g = ol.OracleGraph(n, tr["edge_links"], tr["edge_w"])
        np.testing.assert_array_equal(sy.all_pairs_shortest_paths(board).astype(np.int32), g.dist)
        ell = sy.pack_ell(board)
        for u in range(n):
            nb, w = ell[u] & 0xFFFF, ell[u] >> 16
            real = nb < n
            nodes, wts = g.possible_moves(u, 10**6)
            np.testing.assert_array_equal(nb[real], nodes)
            np.testing.assert_array_equal(w[real], wts)
            assert (nb[~real] == n).all() and (w[~real] == 0xFFFF).all()
            assert (np.diff(nb[real].astype(np.int64)) > 0).all()


def test_pack_rejects_bad_boards():
    with pytest.raises(ValueError):
        sy.make_board(4, [[0, 0]], [1])
    with pytest.raises(ValueError):
        sy.make_board(4, [[0, 7]], [1])
    star = sy.make_board(20, [[0, i] for i in range(1, 19)], np.ones(18))
    with pytest.raises(ValueError):
        sy.pack_ell(star)  # 18 neighbours > ELL width 16
    with pytest.raises(ValueError):
        sy.all_pairs_shortest_paths(sy.make_board(4, [[0, 1]], [1]))  # disconnected
    dup = sy.make_board(3, [[0, 1], [1, 0], [1, 2]], [3, 2, 1])  # parallel edges -> cheapest
    assert (sy.pack_ell(dup)[0, 0] >> 16) == 2


def test_reward_tables_are_the_reference_formulas():
    e, c = sy.reward_tables()
    assert e[0] == 1.0 and e[3] == np.exp(-3.0) and c[0] == 1.0
    assert c[4] == np.exp(-np.log1p(4))
    oe, oc = ol.default_tables(64, 64)
    np.testing.assert_allclose(e[:64], oe, rtol=1e-15)
    np.testing.assert_allclose(c[:64], oc, rtol=1e-15)


def test_capi_library_exports_every_declared_symbol():
    """include/sy_env.h is the contract: every `int sy_*(` / `int32_t|int64_t sy_*(` / `const char *sy_*(` must be exported."""
    with open(os.path.join(ROOT, "include", "sy_env.h")) as f:
        header = f.read()
    declared = set(re.findall(r"^(?:int|int32_t|int64_t|const char \*)\s*(sy_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert declared == set(sy._lib.EXPORTS), declared ^ set(sy._lib.EXPORTS)
    assert os.path.exists(sy.LIB_PATH), "build the engine first: python -m student_mechanism_design_amd.build"
    lib = C.CDLL(sy.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.sy_abi_version.restype = C.c_int
    assert lib.sy_abi_version() == sy._lib.ABI_VERSION


def test_capi_argument_validation_without_gpu():
    """Pure host-side checks of the ABI (no kernel is launched)."""
    lib = sy._lib.load()
    h = C.c_void_p()
    bad = [
        sy._lib.EnvConfig(0, 10, 2, 10, 250, 1, 16, 0, 0, 0, 1, 0, 0),     # no envs
        sy._lib.EnvConfig(4, 10, 8, 10, 250, 1, 16, 0, 0, 0, 1, 0, 0),     # too many police
        sy._lib.EnvConfig(4, 10, 2, 10, 250, 1, 10, 0, 0, 0, 1, 0, 0),     # stride not multiple of 16
        sy._lib.EnvConfig(4, 2000, 2, 10, 250, 1, 2000, 0, 0, 0, 1, 0, 0),  # too many nodes
        sy._lib.EnvConfig(4, 10, 2, 10, 250, 0, 16, 0, 0, 0, 1, 0, 0),     # no graphs
        sy._lib.EnvConfig(4, 10, 2, 10, 250, 1, 16, 0, 0, 0, 1, 17, 0),    # waves per block
        sy._lib.EnvConfig(4, 10, 2, 10, 250, 1, 16, 0, 0, 0, 1, 9, 0),     # odd block sizes use 1.5 waves/episode: <= 7
    ]
    for cfg in bad:
        assert lib.sy_env_create(C.byref(cfg), C.byref(h)) == -1
        assert lib.sy_last_error()
    ok = sy._lib.EnvConfig(4096, 200, 4, 20, 250, 1, 208, 5, 0, 0, 1, 0, 0)
    assert lib.sy_env_create(C.byref(ok), C.byref(h)) == 0
    wpb, blocks, lds = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.sy_env_launch_info(h, C.byref(wpb), C.byref(blocks), C.byref(lds)) == 0
    assert wpb.value * blocks.value >= 4096 and 0 < lds.value <= 160 * 1024
    assert wpb.value == 16 and blocks.value == 256        # BASELINE configs[1]: one 16-episode block per CU
    # explicit block sizes are honoured; a board too big for 16 episodes per block gets fewer (LDS is the limit)
    for want, cfgv in ((12, (4096, 200, 4, 20, 250, 1, 208, 5, 0, 0, 1, 12, 0)), (7, (64, 70, 4, 9, 250, 1, 80, 2, 0, 0, 1, 7, 0))):
        h2 = C.c_void_p()
        assert lib.sy_env_create(C.byref(sy._lib.EnvConfig(*cfgv)), C.byref(h2)) == 0
        assert lib.sy_env_launch_info(h2, C.byref(wpb), C.byref(blocks), C.byref(lds)) == 0 and wpb.value == want
        assert lib.sy_env_destroy(h2) == 0
    h3 = C.c_void_p()
    assert lib.sy_env_create(C.byref(sy._lib.EnvConfig(64, 520, 2, 6, 250, 1, 528, 6, 0, 0, 1, 0, 0)), C.byref(h3)) == 0
    assert lib.sy_env_launch_info(h3, C.byref(wpb), C.byref(blocks), C.byref(lds)) == 0
    assert 1 <= wpb.value < 16 and lds.value <= 160 * 1024 and (wpb.value <= 7 or wpb.value % 2 == 0)
    assert lib.sy_env_destroy(h3) == 0
    assert lib.sy_env_step(h, None, None) == -2       # call order error, not a crash
    assert lib.sy_env_rollout(h, 4, None, None) == -2
    assert lib.sy_env_destroy(h) == 0
    assert lib.sy_action_mask_dense(None, None, None, 3, None, None, 1, None, None) == -1
    assert lib.sy_belief_update(None, None, 3, 16, None, None, 0, None, 1, None) == -1


def test_capi_rejects_batches_whose_rows_overflow_32_bit_cursors():
    """Kernel cursors are 32-bit byte offsets: B * NS * 4 (one belief row of the batch) must stay below 4 GiB."""
    lib = sy._lib.load()
    h = C.c_void_p()
    too_big = sy._lib.EnvConfig(1 << 21, 1024, 2, 10, 250, 1, 1024, 0, 0, 0, 1, 0, 0)   # 2 Mi envs x 1024 nodes x 4 B = 8 GiB
    assert lib.sy_env_create(C.byref(too_big), C.byref(h)) == -1
    assert b"4 GiB" in lib.sy_last_error()
    assert lib.sy_env_bind_status(None, None) == -1
    args = sy._lib.ReturnsArgs()
    assert lib.sy_returns_advantages(C.byref(args), None) == -1       # null pointers are refused, nothing launches


def test_rollout_buffer_validation_fires_before_the_abi_sees_a_pointer():
    """a record from alloc_rollout(T') with T' < T must not reach the kernel (it would write past it)."""
    import types
    import torch
    from student_mechanism_design_amd.env import BatchedScotlandYardEnv, make_rollout_record, record_words
    B, A, NS = 6, 3, 16
    fake = types.SimpleNamespace(B=B, A=A, NS=NS, lib=sy._lib.load(), device=torch.device("cpu"), _belief=object())
    check = BatchedScotlandYardEnv._check_rollout_buffers
    good = make_rollout_record(8, B, A, NS, record_words(A), "cpu", log_prob=True)
    check(fake, good, 8)
    check(fake, good, 8, need_log_prob=True)
    with pytest.raises(ValueError, match="shape"):
        check(fake, good, 9)                                            # too few rows
    with pytest.raises(ValueError, match="log_prob"):
        check(fake, make_rollout_record(8, B, A, NS, record_words(A), "cpu"), 8, need_log_prob=True)
    bad = dict(good)
    bad["mask"] = good["mask"][:, :, :, : NS - 1]
    with pytest.raises(ValueError, match="mask"):
        check(fake, bad, 8)                                             # wrong trailing shape
    bad = dict(good)
    bad["belief"] = good["belief"].double()
    with pytest.raises(ValueError, match="belief"):
        check(fake, bad, 8)                                             # wrong dtype
    bad = dict(good)
    bad["record"] = torch.zeros((8, B, 2 * record_words(A)), dtype=torch.int32)[:, :, ::2]
    with pytest.raises(ValueError, match="contiguous"):
        check(fake, bad, 8)
    bad = dict(good)
    bad["record"] = None
    with pytest.raises(ValueError, match="record"):
        check(fake, bad, 8)
    fake_other = types.SimpleNamespace(B=B, A=A, NS=NS, lib=fake.lib, device=torch.device("cuda", 0), _belief=object())
    with pytest.raises(ValueError, match="lives on"):
        check(fake_other, good, 8)                                      # wrong device
    fake_nobelief = types.SimpleNamespace(B=B, A=A, NS=NS, lib=fake.lib, device=torch.device("cpu"), _belief=None)
    with pytest.raises(ValueError, match="no belief"):
        check(fake_nobelief, good, 8)


def test_reset_epoch_key_schedule():
    from student_mechanism_design_amd.env import _splitmix64
    keys = {_splitmix64((7 + 0xD1B54A32D192ED03 * e) & (2**64 - 1)) for e in range(1, 200)}
    assert len(keys) == 199 and 7 not in keys                       # distinct keys per epoch, none equal to the seed
    assert _splitmix64(0) == 0xE220A8397B1DCDAF                      # SplitMix64 known answer (first output for state 0)


def test_sample_board_redraws_boards_wider_than_the_ell():
    """pack_ell raises above 16 neighbours but the sampler never redrew such a board."""
    from student_mechanism_design_amd import graph as G
    calls = {"n": 0}
    real = G._sample_board_once

    def star_first(n, e, cap, rng):
        calls["n"] += 1
        if calls["n"] == 1:         # a 20-leaf star: degree 20 > 16
            return G.make_board(21, np.array([(0, i) for i in range(1, 21)], dtype=np.int32), np.ones(20, dtype=np.int64))
        return real(n, e, cap, rng)

    G._sample_board_once = star_first
    try:
        b = G.sample_board(21, 30, rng=np.random.default_rng(0))
    finally:
        G._sample_board_once = real
    assert calls["n"] == 2 and G.max_degree(b) <= 16
    G.pack_ell(b)


def test_host_board_sampler_matches_the_reference_sampler_statistics():
    """VERDICT r1 #7.  tests/golden/sampler_stats.json holds what the UNMODIFIED ConnectedGraph.sample
    (graph_layout.py:9-80) realises for (15, 20), (50, 110), (100, 190), (200, 400): degree / tree-degree / max-degree /
    weight histograms and realised edge counts (at (50, 110) the degree cap stops the sampler at 99-105 edges).  The
    host sampler draws from its own Generator, so it is held to them within sampling error."""
    import json
    from tests.helpers import GOLDEN, assert_sampler_stats_close, sampler_stats
    with open(os.path.join(GOLDEN, "sampler_stats.json")) as f:
        g = json.load(f)
    assert [(c["nodes"], c["edges_requested"]) for c in g["configs"]] == [(15, 20), (50, 110), (100, 190), (200, 400)]
    for c in g["configs"]:
        n, e = c["nodes"], c["edges_requested"]
        rng = np.random.default_rng(77 + n)
        boards = [sy.sample_board(n, e, rng=rng) for _ in range(min(c["boards"], 300))]
        got = sampler_stats(n, [b.edge_links for b in boards], [b.edges for b in boards])
        assert_sampler_stats_close(c, got, what=f"host sampler N={n} E={e}")


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    b = sy.sample_board(10, 14, rng=np.random.default_rng(0))
    with pytest.raises(sy.EngineError):
        sy.BatchedScotlandYardEnv(4, [b], 2, 10, np.full(11, 0.5))
    with pytest.raises(sy.EngineError):
        sy.compute_action_mask(np.ones((2, 2)) - np.eye(2), 0, 1.0)
    with pytest.raises(sy.EngineError):
        sy.DeviceBeliefTracker(3, np.ones((3, 3)) - np.eye(3))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "student_mechanism_design_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert "oracle_lib" not in src and "sy_oracle" not in src and "from oracle" not in src, fn


def test_oracle_batch_engine_is_self_consistent():
    """The oracle's batched engine (the GPU checker): rollout == manual stepping of its own actions,
    and OpenMP threads do not change results."""
    boards = sy.sample_board_pool(2, 30, 50, seed=5)
    graphs = [ol.OracleGraph(30, b.edge_links, b.edges.astype(np.int32)) for b in boards]
    eg = np.repeat([0, 1], 8).astype(np.int32)
    w = np.linspace(0.2, 0.8, 11)
    a = ol.OracleBatch(graphs, eg, 16, 3, 9, weights=w, reveal_interval=5, node_stride=32)
    b = ol.OracleBatch(graphs, eg, 16, 3, 9, weights=w, reveal_interval=5, node_stride=32, threads=4)
    a.reset(seed=9)
    b.reset(seed=9)
    ra = a.rollout(60)
    rb = b.rollout(60)
    for k in ra:
        np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)
    c = ol.OracleBatch(graphs, eg, 16, 3, 9, weights=w, reveal_interval=5, node_stride=32)
    c.reset(seed=9)
    for s in range(60):
        np.testing.assert_array_equal(c.pos, ra["pos"][s])
        np.testing.assert_array_equal(c.mask, ra["mask"][s])
        c.step(ra["action"][s])
        np.testing.assert_array_equal(c.reward, ra["reward"][s])
    assert (ra["terminated"] | ra["truncated"]).any()
    # starts are distinct and in range after every (auto-)reset
    srt = np.sort(ra["pos"], axis=-1)
    assert (np.diff(srt, axis=-1) > 0).all() and srt.min() >= 0 and srt.max() < 30


def test_scripts_compile():
    """tools/, examples/, oracle/ and the root scripts are valid Python (they only run on the GPU box or in the
    build container, so nothing else in the CPU suite would notice a typo)."""
    import glob
    import py_compile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "examples", "*.py")) +
                   glob.glob(os.path.join(root, "oracle", "*.py")) + [os.path.join(root, "bench.py"),
                                                                      os.path.join(root, "__graft_entry__.py")])
    assert len(files) >= 10
    for f in files:
        py_compile.compile(f, doraise=True, cfile=os.path.join("/tmp", "sy_pyc_" + os.path.basename(f) + "c"))


def test_optional_torchrl_registration_is_import_guarded():
    """the torchrl EnvBase registration is optional.  torchrl / tensordict are absent offline, so the module
    must import cleanly and `make_torchrl_env` must fail with an ImportError that says what to install (parity unpinned)."""
    import student_mechanism_design_amd.torchrl_env as t
    if t.torchrl_available():
        pytest.skip("torchrl present: the wrapper itself needs a GPU engine to construct")
    with pytest.raises(ImportError, match="torchrl"):
        t.make_torchrl_env(None)


def _host_only_env(N, P, max_degree, wpb=0, max_t=250, B=4096):
    """An engine handle configured on the host only (fake, never dereferenced device pointers; nothing is launched)."""
    lib = sy._lib.load()
    h = C.c_void_p()
    NS = (N + 15) // 16 * 16
    assert lib.sy_env_create(C.byref(sy._lib.EnvConfig(B, N, P, 20, max_t, 1, NS, 5, 0, 0, 1, wpb, 0)), C.byref(h)) == 0, \
        lib.sy_last_error()
    fake = C.c_void_p(0x10000)
    assert lib.sy_env_set_graph_pool(h, fake, fake, fake, fake, max_degree) == 0
    return lib, h


def _kernel_name(lib, h, record=True):
    buf = C.create_string_buffer(128)
    assert lib.sy_env_rollout_kernel_name(h, 1 if record else 0, buf, len(buf)) == 0
    return buf.value.decode()


def test_library_names_the_instance_the_launcher_picks():
    """sy_env_rollout_kernel_name reads the plan the launcher itself uses (csrc/sy_dispatch.hip::plan_rollout): the
    roofline object of the bench line names the kernel that really runs (rocprofv3's kernel trace is compared with it)."""
    def name(N, P, wpb, record, policy, max_degree, hidden=64, max_t=250):
        lib, h = _host_only_env(N, P, max_degree, wpb=wpb, max_t=max_t)
        try:
            if policy:
                fake = 0x10000
                w = sy._lib.MappoWeights(*([fake] * 10))
                assert lib.sy_env_set_policy(h, C.byref(w), hidden) == 0, lib.sy_last_error()
            return _kernel_name(lib, h, record)
        finally:
            lib.sy_env_destroy(h)
    # headline: 5 agents, rows of at most 12 neighbours -> half-wave scan with 2 columns per lane
    assert name(200, 4, 16, True, False, 10) == "sy::rollout3_kernel<4,true,4,false,2>"
    assert name(200, 4, 16, True, False, 13) == "sy::rollout3_kernel<4,true,4,false,0>"      # wider rows: paired scan
    assert name(200, 4, 16, False, False, 8) == "sy::rollout3_kernel<4,false,4,false,2>"
    # 7 agents: 4 columns per agent and pass -> 3 columns per lane up to rows of 12, 4 up to 16 (boards of 129..256 nodes)
    assert name(200, 6, 16, True, False, 10) == "sy::rollout3_kernel<4,true,6,false,3>"
    assert name(200, 6, 16, True, False, 16) == "sy::rollout3_kernel<4,true,6,false,4>"
    assert name(199, 5, 16, True, False, 10) == "sy::rollout3_kernel<4,true,5,false,2>"
    assert name(199, 5, 16, True, False, 16) == "sy::rollout3_kernel<4,true,5,false,0>"
    assert name(48, 6, 16, True, False, 9) == "sy::rollout3_kernel<1,true,6,false,0>"         # small boards: paired scan
    # odd block sizes / boards of more than 256 nodes: the round-1 kernels
    assert name(200, 4, 7, True, False, 10) == "sy::rollout_kernel<4,true,4>"
    assert name(520, 2, 0, True, False, 10, ) == "sy::rollout2_kernel<16,true,2,false>"
    # learned policy in the kernel
    assert name(200, 4, 16, True, True, 8) == "sy::rollout3_kernel<4,true,4,true,2>"
    assert name(200, 3, 16, True, True, 8) == "sy::rollout3_kernel<4,true,0,true,0>"


def test_policy_limits_follow_the_instance_that_runs():
    """sy_env_set_policy and the launcher read ONE plan.  With max_timestep >= 2^20 - 2 the pipeline kernel is
    not eligible (its ring packs the timestep into 20 bits), the round-1 kernel runs the policy with its fixed 2304-byte
    scratch, and a hidden size of 128 would not fit: it must be refused, not accepted with overlapping slices."""
    fake = 0x10000
    w = sy._lib.MappoWeights(*([fake] * 10))
    lib, h = _host_only_env(200, 4, 8, max_t=1 << 20)
    try:
        assert lib.sy_env_set_policy(h, C.byref(w), 128) == -1 and b"at most 64" in lib.sy_last_error()
        assert lib.sy_env_set_policy(h, C.byref(w), 64) == 0
        assert _kernel_name(lib, h) == "sy::rollout2_kernel<4,true,4,true>"
    finally:
        lib.sy_env_destroy(h)
    lib, h = _host_only_env(200, 4, 8)
    try:
        assert lib.sy_env_set_policy(h, C.byref(w), 128) == 0
        assert _kernel_name(lib, h) == "sy::rollout3_kernel<4,true,4,true,2>"
        assert lib.sy_env_set_policy(h, C.byref(w), 132) == -1
        assert lib.sy_env_set_policy(h, None, 0) == 0 and _kernel_name(lib, h) == "sy::rollout3_kernel<4,true,4,false,2>"
    finally:
        lib.sy_env_destroy(h)
    # odd block sizes have no policy instance; a handle without a board pool cannot be planned yet
    lib, h = _host_only_env(200, 4, 8, wpb=7)
    try:
        assert lib.sy_env_set_policy(h, C.byref(w), 64) == -1
    finally:
        lib.sy_env_destroy(h)
    h2 = C.c_void_p()
    assert lib.sy_env_create(C.byref(sy._lib.EnvConfig(64, 200, 4, 20, 250, 1, 208, 5, 0, 0, 1, 0, 0)), C.byref(h2)) == 0
    assert lib.sy_env_set_policy(h2, C.byref(w), 64) == -2
    lib.sy_env_destroy(h2)


def test_bench_starts_its_own_ranks_without_touching_the_gpu():
    """`python bench.py --gpus N` without a launcher: the parent starts N children with the environment
    torch.distributed.run would give them, relays rank 0's line and returns the worst exit code — and it does so before
    importing torch or touching the GPU (the parent of GPU work must never have initialised the GPU itself)."""
    import bench

    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, stderr=None):
            self.cmd, self.env, self.rank = cmd, env, int(env["RANK"])
            self.returncode = None
            self.terminated = False
            started.append(self)

        def communicate(self, timeout=None):
            self.returncode = 0
            return (b'{"n_gpus": 3}\n', None)

        def wait(self, timeout=None):
            self.returncode = 3 if self.rank == 2 else 0
            return self.returncode

        def terminate(self):
            self.terminated = True

        def kill(self):
            self.terminated = True

    rc = bench.spawn_ranks(3, ["--gpus", "3", "--steps", "2"], popen=FakeProc)
    assert rc == 3                                                  # the worst child
    assert [p.rank for p in started] == [0, 1, 2]
    ports = {p.env["MASTER_PORT"] for p in started}
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536
    for p in started:
        assert p.env["WORLD_SIZE"] == "3" and p.env["LOCAL_RANK"] == p.env["RANK"] and p.env["MASTER_ADDR"] == "127.0.0.1"
        assert p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert p.cmd[0] == sys.executable and p.cmd[1].endswith("bench.py") and p.cmd[2:] == ["--gpus", "3", "--steps", "2"]
    # the real thing, as a subprocess, on a machine without a GPU: both ranks fail at once (no device), the parent must
    # report that with a non-zero exit code and must not have imported torch itself
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would really run")
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n"
            "    print('RC', e.code, 'TORCH', 'torch' in sys.modules)\n" % os.path.join(ROOT, "bench.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "TORCH False" in out.stdout and "RC 0" not in out.stdout, (out.stdout, out.stderr[-500:])


def test_ppo_gradient_abi_sizes_and_argument_checks():
    """sy_ppo_pack / sy_mappo_ppo_grad (include/sy_env.h): the layout helpers agree with the host mirror, and bad arguments are
    refused with a message before anything is launched (no GPU here: a launch would fail differently)."""
    from student_mechanism_design_amd.update import MappoUpdater
    L = sy._lib
    lib = L.load()
    for N, H in ((200, 64), (15, 8), (100, 128), (1024, 128)):
        S = int(lib.sy_ppo_slab_floats(N, H))
        assert S == MappoUpdater._slab_floats(N, H) == 2 * N * H + H + ((max(N, H) + 3) & ~3) + 8
        assert int(lib.sy_ppo_scratch_floats(5, N, H)) % (6 * S) == 0
    assert int(lib.sy_ppo_image_bytes(5, 1000)) == 1000 * (16 + 16 * 5 + 8)
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    pack = L.PpoPackArgs(p, 32, p, p, p, None, 0, 100, 8, p, 4, p, int(lib.sy_ppo_image_bytes(5, 100)), 0, 0)
    for field, value, needle in (("image", None, "null"), ("num_police", 0, "bad sizes"), ("record_words", 8, "record_words"),
                                 ("image_bytes", 16, "image too small"), ("shuffle_domain", 50, "shuffle_domain")):
        a = L.PpoPackArgs.from_buffer_copy(pack)
        setattr(a, field, value)
        assert lib.sy_ppo_pack(C.byref(a), None) == -1
        assert needle in lib.sy_last_error().decode(), (field, lib.sy_last_error())
    a = L.PpoPackArgs.from_buffer_copy(pack)
    a.rows, a.shuffle_domain = p, 100
    assert lib.sy_ppo_pack(C.byref(a), None) == -1 and "not both" in lib.sy_last_error().decode()
    grad = L.PpoArgs(p, 100, 0, None, 100, p, 4, 200, 64, p, 0.2, 0.5, p, int(lib.sy_ppo_scratch_floats(5, 200, 64)), p, None, None, None,
                     3e-4, 0.9, 0.999, 1e-8)
    for field, value, needle in (("params", None, "null"), ("hidden", 66, "multiple of 4"), ("hidden", 132, "multiple of 4"),
                                 ("scratch_floats", 10, "scratch too small"), ("num_rows", 101, "past the image"),
                                 ("num_nodes", 1, "bad sizes"), ("adam_m", p, "go together")):
        a = L.PpoArgs.from_buffer_copy(grad)
        setattr(a, field, value)
        assert lib.sy_mappo_ppo_grad(C.byref(a), None) == -1
        assert needle in lib.sy_last_error().decode(), (field, lib.sy_last_error())
    a = L.PpoArgs.from_buffer_copy(grad)
    a.adam_m, a.adam_v, a.adam_step, a.lr = p, p, p, 0.0
    assert lib.sy_mappo_ppo_grad(C.byref(a), None) == -1 and "Adam constants" in lib.sy_last_error().decode()


def test_belief_layout_is_a_conflict_free_relabelling():
    """graph.belief_layout: every node keeps exactly its ELL row's neighbours (another visit order, other scratch entries),
    entries are distinct, never the zero entry, inside the scratch; under the MI355X bank model the gathers of a diffusion
    step lose their bank conflicts (node order: ~175 extra LDS cycles per pair-step at 200 nodes) and the scatter of the
    lanes' own entries stays (nearly) conflict-free."""
    from student_mechanism_design_amd import graph as G
    for n, e, seed in ((200, 400, 0), (199, 390, 3), (100, 170, 1), (40, 75, 2), (256, 500, 4), (15, 20, 5)):
        boards = sy.sample_board_pool(2, n, e, seed=seed)
        pool = G.pack_pool(boards)
        N, NS = pool.num_nodes, pool.node_stride
        for g in range(2):
            ell = pool.ell[g]
            slot, gather = G.belief_layout(ell, N, NS)
            assert slot.dtype == np.uint16 and gather.dtype == np.uint16 and gather.shape == (N, 16) and slot.shape == (NS,)
            own = [int(x) for x in slot[:N]]
            assert len(set(own)) == N and N not in own and max(own) < NS + 16 and (slot[N:] == N).all()
            back = {s * 8: u for u, s in enumerate(own)}
            back[N * 8] = N
            for v in range(N):
                assert sorted(int(x) for x in (ell[v] & 0xFFFF)) == sorted(back[int(x)] for x in gather[v]), v
            ident_slot = np.arange(NS, dtype=np.uint16)
            ident_slot[N:] = N
            ident = ((ell & 0xFFFF).astype(np.uint32) * 8).astype(np.uint16)
            before = G.belief_bank_conflicts(ident, ident_slot, N, NS)["gather_extra_cycles"]
            after = G.belief_bank_conflicts(gather, slot, N, NS)
            assert after["gather_extra_cycles"] <= max(2, before // 20), (n, before, after)
            assert after["store_extra_cycles"] <= 16, after
    assert G.belief_layout(np.zeros((300, 16), np.uint32), 300, 304) is None      # the older kernels' boards: no layout
