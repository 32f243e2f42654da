"""CPU: the host logic of the launch-plan replay (engine/plan.py) on synthetic timelines -- which cross-stream edges the memory accesses of
a recorded step imply, and the allocator-block map that resolves pointer arguments.  (The replay itself is tested on the GPU:
tests/test_plan_gpu.py.)"""
from mgnet_amd.engine.plan import BREAK, LAUNCH, RECORD, WAIT, _Blocks, derive_schedule

A, B, C, D = (0x1000, 0x2000), (0x3000, 0x4000), (0x5000, 0x6000), (0x7000, 0x8000)
MAIN, S1, S2 = 11, 22, 33


def item(node, stream, reads=(), writes=(), kind=0):
    return dict(kind=kind, node=node, stream=stream, reads=list(reads), writes=list(writes), name=f"k{node}")


def ordered_before(ops, first, second):
    """is launch `first` ordered before launch `second` by stream order and record / wait pairs?  (reachability over the op list)"""
    # happens-before frontier per stream: set of launches known complete when the stream reaches a point
    done = {}                 # stream -> set of nodes
    at_event = {}
    for typ, a, st in ops:
        cur = done.setdefault(st, set())
        if typ == LAUNCH:
            if a == second:
                return first in cur
            cur.add(a)
        elif typ == RECORD:
            at_event[a] = set(cur)
        elif typ == WAIT:
            cur |= at_event.get(a, set())
    raise AssertionError("second launch not found")


def test_same_stream_needs_no_events():
    ops, n_ev, n_cross, ns, _ = derive_schedule([item(0, MAIN, writes=[A]), item(1, MAIN, reads=[A], writes=[B]), item(2, MAIN, reads=[B])], MAIN)
    assert n_cross == 0 and ns == 1 and [o[0] for o in ops] == [RECORD, LAUNCH, LAUNCH, LAUNCH]


def test_raw_war_waw_across_streams_and_read_read_is_free():
    items = [item(0, MAIN, writes=[A]),                 # producer
             item(1, S1, reads=[A], writes=[B]),        # RAW on A -> must wait for 0
             item(2, S2, reads=[A], writes=[C]),        # RAW on A; read-read with 1: no edge between 1 and 2
             item(3, MAIN, writes=[A]),                 # WAR: must wait for the readers 1 and 2
             item(4, S1, writes=[C])]                   # WAW with 2
    ops, n_ev, n_cross, ns, _ = derive_schedule(items, MAIN)
    assert ns == 3
    assert ordered_before(ops, 0, 1) and ordered_before(ops, 0, 2)
    assert ordered_before(ops, 1, 3) and ordered_before(ops, 2, 3)
    assert ordered_before(ops, 2, 4)
    assert not ordered_before(ops, 1, 2)                # two readers of A on different streams stay concurrent
    # every wait refers to an event recorded earlier in the list
    seen = set()
    for typ, a, st in ops:
        if typ == RECORD:
            seen.add(a)
        if typ == WAIT:
            assert a in seen


def test_block_reuse_across_streams_is_ordered():
    """a block freed by stream S1's tensor and handed to a later tensor of MAIN (the allocator did that after the HOST saw S1's work
    complete -- nothing a replay repeats): the overlapping addresses order the two launches"""
    big, part = (0x10000, 0x20000), (0x14000, 0x15000)
    items = [item(0, S1, reads=[big]), item(1, MAIN, writes=[part])]
    ops, *_ = derive_schedule(items, MAIN)
    assert ordered_before(ops, 0, 1)


def test_transitive_edges_are_not_duplicated():
    items = [item(0, S1, writes=[A]), item(1, S1, writes=[B]), item(2, MAIN, reads=[B]), item(3, MAIN, reads=[A])]
    ops, n_ev, n_cross, ns, _ = derive_schedule(items, MAIN)
    assert n_cross == 1                                  # waiting for launch 1 covers launch 0 (same stream, earlier)
    assert ordered_before(ops, 0, 3) and ordered_before(ops, 1, 2)


def test_closures_become_breaks_and_streams_join_at_both_ends():
    items = [item(0, S1, writes=[A]), item(None, MAIN, reads=[A], writes=[B], kind=1), item(2, MAIN, reads=[B])]
    ops, n_ev, n_cross, ns, _ = derive_schedule(items, MAIN)
    kinds = [o[0] for o in ops]
    assert kinds.count(BREAK) == 1 and kinds[0] == RECORD and ops[1] == (WAIT, ops[0][1], S1)      # the side stream starts behind main
    assert ops[-1][0] == WAIT and ops[-1][2] == MAIN and ops[-2] == (RECORD, ops[-1][1], S1)         # main ends behind the side stream
    i_break = kinds.index(BREAK)
    assert (WAIT, 0, MAIN) in ops[:i_break]              # the closure's stream waited for launch 0 before the host acts


def test_serial_mode_orders_everything():
    items = [item(0, S1, writes=[A]), item(1, S2, writes=[B]), item(2, MAIN, writes=[C])]
    ops, *_ = derive_schedule(items, MAIN, serial=True)
    assert ordered_before(ops, 0, 1) and ordered_before(ops, 1, 2)


def test_block_map_newest_allocation_wins():
    b = _Blocks()
    b.add(0x1000, 0x9000)                                # a large tensor ...
    assert b.find(0x4000) == (0x1000, 0x9000)
    b.add(0x1000, 0x2000)                                # ... freed, its block re-split into smaller ones
    b.add(0x3000, 0x5000)
    assert b.find(0x1800) == (0x1000, 0x2000) and b.find(0x4000) == (0x3000, 0x5000)
    assert b.find(0x2800) is None and b.find(0x8000) is None      # the rest of the old block is nobody's now
    b.add(0x4000, 0x6000)                                # overlapping the previous one: it must be dead
    assert b.find(0x3800) is None and b.find(0x4800) == (0x4000, 0x6000)
    b.add(0x4000, 0x6000)                                # the same block again: unchanged
    assert b.find(0x5fff) == (0x4000, 0x6000) and b.find(0x6000) is None


def test_static_batch_copy_keeps_the_view_structure_and_refills():
    """Trainer.run_step_planned's input side: plan-owned copies of a collated batch are views of as many buffers as the batch had, a batch
    collated the same way refills them with one copy per buffer, a batch of separate tensors entry by entry"""
    import torch
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine.trainer import Trainer
    t = Trainer.__new__(Trainer)
    b = synthetic_batch(2, 32, 64, "cpu")
    static, bases, layout = t._static_copy(b)
    n_buffers = len({id(v._base) for d in b for v in d.values() if isinstance(v, torch.Tensor) and v._base is not None})
    assert len(bases) == n_buffers
    for d, o in zip(b, static):
        for k, v in d.items():
            if isinstance(v, torch.Tensor):
                assert torch.equal(v, o[k]) and v.data_ptr() != o[k].data_ptr() and o[k].stride() == v.stride(), k
    t._static_bases, t._static_layout, t._plan_inputs = bases, layout, static
    same = synthetic_batch(2, 32, 64, "cpu", seed=5)
    assert t._layout(same)[1] == layout
    loose = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()} for d in synthetic_batch(2, 32, 64, "cpu", seed=6)]
    assert t._layout(loose)[1] != layout
    for new in (same, loose):
        t._refill_static(new)
        for d, o in zip(new, static):
            for k, v in d.items():
                if isinstance(v, torch.Tensor):
                    assert torch.equal(v, o[k]), k
    assert Trainer._batch_signature(b) == Trainer._batch_signature(loose) != Trainer._batch_signature(synthetic_batch(2, 32, 96, "cpu"))
    # what the dataset mapper adds per sample (file names, ids) is not part of the signature: the plan is replayed for them
    named = [dict(d, file_name=f"f{j}.png", image_id=j) for j, d in enumerate(b)]
    assert Trainer._batch_signature(named) == Trainer._batch_signature(b)



def test_critical_path_tool_reads_the_schedule_edges_and_walks_the_last_chain():
    """tools/critical_path.py: the dependency edges it reconstructs from the op list (same-stream order + record / wait pairs) and the
    chain it walks back from the kernel that ends last, on a synthetic two-stream step with a host-issued torch op in the middle"""
    import importlib.util
    import os
    import types

    import numpy as np
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("critical_path", os.path.join(root, "tools", "critical_path.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    items = [item(0, MAIN, writes=[A]),              # producer
             item(1, S1, reads=[A], writes=[B]),     # side stream: waits for 0
             item(2, MAIN, writes=[C]),              # independent of 1
             item(-1, MAIN, reads=[C], writes=[C], kind=1),   # a torch op between plan segments (no events of its own)
             item(3, MAIN, reads=[B, C], writes=[D])]         # joins both: waits for 1 (event) and follows the closure
    items[3]["name"] = "add.Tensor"
    ops, n_ev, n_cross, ns, _ = derive_schedule(items, MAIN)
    plan = types.SimpleNamespace(ops=ops, items=[dict(kind=it["kind"], node=it["node"], stream=it["stream"], name=it["name"]) for it in items],
                                 main=types.SimpleNamespace(cuda_stream=MAIN), handle=None)
    preds = cp.schedule_edges(plan)
    assert preds[0] == [] and preds[1] == [0] and preds[2] == [0] and preds[3] == [2] and sorted(preds[4]) == [1, 3]
    # node times (ms): 0: 0-1, 1: 1.1-5 (the long one), 2: 1.05-2, 3: 5.2-6
    begins = [np.array([0.0, 1.1, 1.05, 5.2])]
    ends = [np.array([1.0, 5.0, 2.0, 6.0])]
    cp.plan_grid = lambda *a: ""
    out = []
    res = cp.analyse(plan, begins, ends, cp.FAMILIES, out)
    txt = "\n".join(out)
    assert abs(res["trace_span_ms"] - 6.0) < 1e-9 and res["kernels"] == 4
    chain = txt[txt.index("(b) the dependency chain"):]
    rows = [ln.split() for ln in chain.splitlines() if ln[:9].strip().replace(".", "").isdigit()]
    names = [r[-1] if not r[-1].startswith("(") else r[-2] for r in rows]
    assert names == ["k0", "k1", "k3"], (names, chain)       # 0 -> 1 (cross-stream) -> 3; k2 and the torch op are off the chain
    assert abs(res["no_kernel_ms"] - (0.05 + 0.2)) < 1e-6        # 1.0-1.05 and 5.0-5.2
