"""Sharded parse + merge (semantics of PfParser::operator+= / merge_pfp): N local parses merged
must equal the single parse bit for bit, hence also every later file.
* CPU: through tests/emu (single process) and through pfbwt_dist with torch.distributed gloo, world size 2.
* GPU: N contexts on one MI355X (single process)."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import pytest
from pfp_testlib import EMU_SO, ROOT, compare, oracle_run


def synth(seed, L, H, nruns=(0, 0, 0, 0)):
    lib = C.CDLL(os.path.join(ROOT, "pfbwt-f_amd", "lib", "libpfpsynth.so"))
    lib.pfp_synth_haplotype.argtypes = [C.c_uint64] * 7 + [C.c_void_p]
    out = []
    for h in range(H):
        a = np.empty(L, np.uint8); lib.pfp_synth_haplotype(seed, L, h, *nruns, a.ctypes.data_as(C.c_void_p)); out.append(a.tobytes())
    return out


def sharded_single_process(factory, seqs, shards, w, p, U, mode="context", compact=False):
    """mode: "context" -- shard r > 0 is parsed with the w 'A's of its left neighbour (the multi-GPU convention);
    "standalone" -- every shard is parsed on its own, the merge re-tests the first w windows (PfParser::operator+=, :226-245);
    "loaded" -- every shard is parsed on its own by the ORACLE, saved as .dict / .parse images and loaded (merge_pfp from files)"""
    ctxs, views = [], []
    for r, grp in enumerate(shards):
        c = factory(w=w, p=p, u64=(U == 8), sai=True)
        if mode == "loaded":
            o = oracle_run([seqs[i] for i in grp], w=w, p=p, U=U)
            c.shard_load(o["dict"], o["parse"])
        else:
            if r > 0 and mode == "context":
                c.feed_left_context(w)
            for i in grp:
                c.feed(seqs[i], True)
            c.finalize(shard=(len(shards) % 2 == 1))      # both ways: a shard needs no dictionary sort / ranks of its own
        ctxs.append(c); views.append(c.shard_view())
        assert views[-1].left_context == (w if (r > 0 and mode == "context") else 0)
        if compact:      # what travels between GPUs: dictionary, word starts, phrase ids -- the merge derives phrase ends and last bytes
            views[-1].d_ye = views[-1].d_last = None
    g = factory(w=w, p=p, u64=(U == 8), sai=True)
    sz = g.merge_shards(views)
    res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize}
    res.update(g.parse_get()); g.parse_bwt(); res.update(g.parse_bwt_get())
    b = g.bwt_build(sa=True, rssa=True); res.update(g.bwt_get()); res["r"] = b.r
    for c in ctxs + [g]:
        c.close()
    return res


@pytest.fixture(scope="module")
def emu_factory():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pfbwt-f_amd"), "emu"], check=True, stdout=subprocess.DEVNULL)
    import pfbwt_hip
    return lambda **kw: pfbwt_hip.PfpContext(lib=EMU_SO, **kw)


def test_sharded_merge_emu(emu_factory):
    seqs = synth(5, 4000, 4)
    for w, p, shards in ((10, 100, [[0], [1], [2], [3]]), (4, 7, [[0, 1], [2, 3]]), (4, 7, [[0], [1, 2, 3]])):
        ref = oracle_run(seqs, w=w, p=p, U=4)
        assert compare(sharded_single_process(emu_factory, seqs, shards, w, p, 4), ref, 4) == []
        assert compare(sharded_single_process(emu_factory, seqs, shards, w, p, 4, compact=True), ref, 4) == []


def sharded_c_api(lib, seqs, shards, w, p, U, devices, sa=True):
    """pfp_sharded_* (one process, N ranks): the slices in rank order == the single build"""
    import pfbwt_hip
    sb = pfbwt_hip.ShardedBuild(len(shards), devices=devices, w=w, p=p, u64=(U == 8), lib=lib)
    res = None
    for attempt in range(2):      # the handle is reusable: reset + feed again
        for r, grp in enumerate(shards):
            for i in grp:
                sb.rank(r).feed(seqs[i], True)
        ps, slices = sb.build(sa=sa, rssa=True)
        parts = {"bwt": [], "sa": [], "ssa": [], "esa": []}; r_tot = 0; pos = 0
        for r, (b, beg, rows) in enumerate(slices):
            assert beg == pos; pos += rows
            o = sb.rank(r).bwt_get()
            for k in parts:
                if o.get(k) is not None: parts[k].append(o[k])
            r_tot += b.r
        res = {k: np.concatenate(v) for k, v in parts.items() if v}
        res.update(r=r_tot, n=ps.n, m=ps.m, dwords=ps.dwords, dsize=ps.dsize)
        assert pos == ps.n + 1
        if attempt == 0: sb.reset()
    sb.close()
    return res


def test_sharded_c_api_emu(emu_factory):
    """the C-level sharded build on the CPU interpreter: 3 and 2 ranks (unequal shards), -s -r and -r, two builds per handle; a failing
    rank stops every rank in front of the exchange"""
    import pfbwt_hip
    seqs = synth(5, 4000, 5)
    for w, p, shards, sa in ((10, 100, [[0], [1, 2], [3, 4]], True), (4, 7, [[0, 1, 2, 3], [4]], False)):
        ref = oracle_run(seqs, w=w, p=p, U=8)
        res = sharded_c_api(EMU_SO, seqs, shards, w, p, 8, [0] * len(shards), sa=sa)
        assert compare(res, ref, 8, names=("bwt", "sa", "ssa", "esa") if sa else ("bwt", "ssa", "esa")) == [] and res["r"] == ref["r"]
        assert (res["n"], res["m"], res["dwords"], res["dsize"]) == (ref["n"], ref["m"], ref["dwords"], ref["dsize"])
    sb = pfbwt_hip.ShardedBuild(2, devices=[0, 0], w=10, p=100, lib=EMU_SO)
    sb.rank(0).feed(seqs[0], True); sb.rank(1).feed(b"ACGTRACGT" * 50, True)
    with pytest.raises(pfbwt_hip.PfpError) as e:
        sb.build()
    assert e.value.status == -2 and "rank 1" in str(e.value)
    sb.reset()
    sb.rank(0).feed(seqs[0], True); sb.rank(1).feed(seqs[1], True)
    ps, _ = sb.build(sa=False, rssa=True)
    assert ps.n == 2 * (4000 + 10)
    sb.close()


def seam_trigger_seqs():
    """sequences whose first w windows hold triggers for (w, p) = (4, 3) and (6, 2): a stand-alone parse cannot cut there"""
    rng = np.random.default_rng(17)
    return [bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in (900, 35, 700, 5, 1200, 64, 300)]


def test_shard_finalize_state_emu(emu_factory):
    """pfp_parse_finalize_shard: the context answers pfp_shard_view_get and nothing that needs ranks (PFP_E_STATE)"""
    import pfbwt_hip
    c = emu_factory(w=4, p=7, u64=True, sai=True)
    for s in synth(3, 3000, 2): c.feed(s, True)
    sz = c.finalize(shard=True)
    v = c.shard_view()
    assert v.m == sz.m and v.dwords == sz.dwords and sz.m > 2
    for call in (c.parse_get, c.parse_bwt):
        with pytest.raises(pfbwt_hip.PfpError) as e:
            call()
        assert e.value.status == pfbwt_hip.E_STATE
    c.close()


@pytest.mark.parametrize("mode", ["standalone", "loaded"])
def test_merge_of_standalone_shards_emu(emu_factory, mode):
    """pfp_merge_shards on shards that were parsed WITHOUT knowing their left neighbour (merge_pfp on saved parses,
    src/merge_pfp.cpp:97-113): the seam is re-hashed like PfParser::operator+= (pfparser.hpp:226-245) -- extra junction phrases
    where the first w windows of a shard hold triggers; result == the single parse on every array"""
    seqs = seam_trigger_seqs()
    extra_seen = False
    for w, p, shards in ((4, 3, [[0], [1], [2], [3], [4], [5], [6]]), (6, 2, [[0, 1], [2], [3, 4, 5], [6]]), (10, 100, [[0, 1, 2], [3, 4, 5, 6]])):
        ref = oracle_run(seqs, w=w, p=p, U=8)
        res = sharded_single_process(emu_factory, seqs, shards, w, p, 8, mode=mode)
        assert compare(res, ref, 8) == [], (mode, w, p)
        parts = sum(int(oracle_run([seqs[i] for i in g], w=w, p=p, U=8)["m"]) for g in shards)
        extra_seen |= int(ref["m"]) > parts - (len(shards) - 1)            # more phrases than "every seam fuses two fragments into one"
    assert extra_seen, "no seam of the test inputs needed an extra junction phrase"


def one_phrase_cases():
    """operands with a single phrase (a record shorter than any trigger window allows: nothing of it closes a phrase) in first,
    middle and last position, two of them in a row, every operand a single phrase, and a single operand"""
    rng = np.random.default_rng(23)
    big = [bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in (3000, 2000, 1500)]
    tiny = [bytes(rng.choice(list(b"ACGT"), n).astype(np.uint8)) for n in (30, 12, 1, 0, 25)]
    return [([big[0], tiny[0], big[1]], [[0], [1], [2]]), ([tiny[0], big[0], big[1]], [[0], [1], [2]]), ([big[0], big[1], tiny[1]], [[0], [1], [2]]),
            ([big[0], tiny[0], tiny[1], tiny[2], big[1], tiny[3]], [[0], [1], [2], [3], [4], [5]]), ([tiny[0], tiny[1], tiny[4]], [[0], [1], [2]]),
            ([tiny[4], tiny[3], big[2]], [[0], [1], [2]]), ([big[0], big[1]], [[0, 1]]), ([tiny[0]], [[0]])]


def check_one_phrase_operands(factory, modes):
    import pfbwt_hip
    for seqs, shards in one_phrase_cases():
        for w, p in ((10, 100), (4, 3)):
            ref = oracle_run(seqs, w=w, p=p, U=8)
            parts = [int(oracle_run([seqs[i] for i in g], w=w, p=p, U=8)["m"]) for g in shards]
            for mode in modes:
                if int(ref["m"]) < 2:      # "only one dict word total" (pfparser.hpp:390-392): the merge itself must still succeed
                    with pytest.raises(pfbwt_hip.PfpError) as e:
                        sharded_single_process(factory, seqs, shards, w, p, 8, mode=mode)
                    assert e.value.status == pfbwt_hip.E_ONE_WORD, (mode, w, p, parts)
                    continue
                assert compare(sharded_single_process(factory, seqs, shards, w, p, 8, mode=mode), ref, 8) == [], (mode, w, p, parts)


def test_merge_with_one_phrase_operands_emu(emu_factory):
    """ADVICE r2 (merge_pfp.cpp:88): an operand whose parse is ONE phrase is head and tail fragment of its seams at once; the
    reference folds it into the open phrase (PfParser::operator+=, pfparser.hpp:194-263) and so does pfp_merge_shards"""
    check_one_phrase_operands(emu_factory, ("standalone", "loaded", "context"))


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "pfbwt-f_amd", "python"))
from pfp_testlib import EMU_SO, compare, oracle_run
from test_sharded import synth
import pfbwt_hip, pfbwt_dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seqs = synth(7, 3000, 5)
seqs = [seqs[0], seqs[1][:1700], seqs[2][:37], seqs[3], seqs[4][:2211]]      # unequal shards: rank 0 holds four records (one shorter
mine = seqs[:4] if rank == 0 else seqs[4:]                                    # than a phrase), rank 1 a single one
assert world == 2
ctx = pfbwt_hip.PfpContext(lib=EMU_SO, w=6, p=11, u64=False, sai=True)
sz, b, begin, rows = pfbwt_dist.sharded_build(ctx, lambda c: [c.feed(s, True) for s in mine], 6, torch.device("cpu"), sa=True)
o = ctx.bwt_get()
parts = [None] * world
dist.all_gather_object(parts, (begin, rows, int(b.r), o["bwt"], o["sa"]))
ok = 1
if rank == 0:
    parts.sort(key=lambda t: t[0])
    assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize, "r": sum(t[2] for t in parts)}
    res.update(ctx.parse_get()); res.update(ctx.parse_bwt_get())
    res["bwt"] = np.concatenate([t[3] for t in parts]); res["sa"] = np.concatenate([t[4] for t in parts])
    bad = compare(res, oracle_run(seqs, w=6, p=11, U=4), 4, names=("dict", "occ", "parse", "last", "sai", "bwlast", "ilist", "bwsai", "bwt", "sa"))
    ok = 0 if bad else 1
    if bad: print("MISMATCH", bad, flush=True)
# the same build again in -r mode (what bench.py runs by default): run samples stay sliced over the ranks
sz, b, begin, rows = pfbwt_dist.sharded_build(ctx, lambda c: [c.feed(s, True) for s in mine], 6, torch.device("cpu"), sa=False, rssa=True)
o = ctx.bwt_get()
dist.all_gather_object(parts, (begin, rows, int(b.r), o["bwt"], o["ssa"], o["esa"]))
if rank == 0:
    parts.sort(key=lambda t: t[0])
    res = {"r": sum(t[2] for t in parts), "bwt": np.concatenate([t[3] for t in parts]), "ssa": np.concatenate([t[4] for t in parts]), "esa": np.concatenate([t[5] for t in parts])}
    bad = compare(res, oracle_run(seqs, w=6, p=11, U=4), 4, names=("bwt", "ssa", "esa"))
    if bad: ok = 0; print("MISMATCH -r", bad, flush=True)
# a rank whose shard is rejected (invalid character): EVERY rank must raise, nobody may wait in the all-gather
bad = [seqs[4][:500] + b"R" + seqs[4][500:1000]]
try:
    pfbwt_dist.sharded_build(ctx, lambda c: [c.feed(s, True) for s in (bad if rank == 1 else mine)], 6, torch.device("cpu"), sa=True)
    ok = 0; print("rank %d: no error raised" % rank, flush=True)
except pfbwt_hip.PfpError as e:
    if rank != 1 or e.status != pfbwt_hip.E_INVALID_CHAR: ok = 0; print("rank %d: unexpected %r" % (rank, e), flush=True)
except RuntimeError as e:
    if rank != 0: ok = 0; print("rank %d: unexpected %r" % (rank, e), flush=True)
ctx.reset()
t = torch.tensor([ok]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if int(t) == 1 else 1)
'''


def test_sharded_build_gloo_world2(emu_factory, tmp_path):
    """the N > 1 path of bench.py (pfbwt_dist.sharded_build: all-gather of dictionaries, merge, sliced emission), gloo, 2 ranks"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29641",
                         str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]


WORKER4 = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "pfbwt-f_amd", "python"))
from pfp_testlib import EMU_SO, compare, oracle_run
from test_sharded import synth
import pfbwt_hip, pfbwt_dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 4
big = synth(11, 2500, 4)
# rank 1 holds ONE record shorter than any phrase (its parse is a single phrase), rank 2 two records one of which is empty
shards = [[big[0], big[1][:900]], [big[2][:23]], [b"", big[3]], [big[1][900:], big[2][23:800]]]
names = [["a0", "a1"], ["b0"], ["c0", "c1"], ["d0", "d1"]]
seqs = [s for sh in shards for s in sh]
mine = shards[rank]
w, p = 6, 11
ctx = pfbwt_hip.PfpContext(lib=EMU_SO, w=w, p=p, u64=True, sai=True)
ok = 1
for sa, rssa in ((True, False), (False, True)):
    sz, b, begin, rows = pfbwt_dist.sharded_build(ctx, lambda c: [c.feed(s, True) for s in mine], w, torch.device("cpu"), sa=sa, rssa=rssa)
    o = ctx.bwt_get()
    parts = [None] * world
    dist.all_gather_object(parts, (begin, rows, int(b.r), o["bwt"], o["sa"], o["ssa"], o["esa"]))
    if rank == 0:
        parts.sort(key=lambda t: t[0])
        assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize, "r": sum(t[2] for t in parts), "bwt": np.concatenate([t[3] for t in parts])}
        if sa: res["sa"] = np.concatenate([t[4] for t in parts])
        if rssa: res["ssa"] = np.concatenate([t[5] for t in parts]); res["esa"] = np.concatenate([t[6] for t in parts])
        res.update(ctx.parse_get())
        bad = compare(res, oracle_run(seqs, w=w, p=p, U=8), 8, names=("dict", "occ", "parse", "last", "sai", "bwt") + (("sa",) if sa else ()) + (("ssa", "esa") if rssa else ()))
        if bad: ok = 0; print("MISMATCH", sa, rssa, bad, flush=True)
# .docs through the sharded build
st, acc = [], 0
for s in mine:
    st.append(acc); acc += len(s) + w
docs = pfbwt_dist.allgather_docs(list(zip(names[rank], st)), acc)
want, acc = [], 0
for nm, s in zip([n for sh in names for n in sh], seqs):
    want.append((nm, acc)); acc += len(s) + w
if docs != want: ok = 0; print("rank %d: docs %r != %r" % (rank, docs, want), flush=True)
ctx.close()
t = torch.tensor([ok]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if int(t) == 1 else 1)
'''


def test_sharded_build_gloo_world4_with_one_phrase_shard_and_docs(emu_factory, tmp_path):
    """four ranks, one of them with a shard whose parse is a single phrase, one with an empty record; -s and -r; .docs gathered"""
    script = tmp_path / "worker4.py"
    script.write_text(WORKER4)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "29643",
                         str(script), ROOT], capture_output=True, text=True, env=env, timeout=900)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-3000:]


def sliced_emission(factory, seqs, w, p, U, nslices):
    bwt, sa, r = [], [], 0
    for sl in range(nslices):
        c = factory(w=w, p=p, u64=(U == 8), sai=True)
        for s in seqs:
            c.feed(s, True)
        c.finalize(); c.parse_bwt()
        b, beg, rows = c.bwt_build_slice(sl, nslices, sa=True)
        o = c.bwt_get(); c.close()
        assert beg == sum(len(x) for x in bwt) and rows == len(o["bwt"])
        bwt.append(o["bwt"]); sa.append(o["sa"]); r += b.r
    return np.concatenate(bwt), np.concatenate(sa), r


def sliced_samples(factory, seqs, w, p, U, nslices, ref, sa):
    """-r over slices: every slice's ssa / esa pairs, concatenated in slice order, are the reference's .ssa / .esa"""
    parts = {"bwt": [], "ssa": [], "esa": []}
    r = 0
    for sl in range(nslices):
        c = factory(w=w, p=p, u64=(U == 8), sai=True)
        for s in seqs:
            c.feed(s, True)
        c.finalize(); c.parse_bwt()
        b, beg, rows = c.bwt_build_slice(sl, nslices, sa=sa, rssa=True)
        o = c.bwt_get(); c.close()
        assert len(o["ssa"]) == 2 * b.r
        for k in parts:
            parts[k].append(o[k])
        r += b.r
    res = {k: np.concatenate(v) for k, v in parts.items()}
    res["r"] = r
    assert compare(res, ref, U, names=("bwt", "ssa", "esa")) == [], (nslices, sa)


def test_sliced_emission_emu(emu_factory):
    seqs = synth(5, 4000, 3)
    ref = oracle_run(seqs, w=4, p=7, U=4)       # many multi-word groups straddle the slice boundaries
    for ns in (3,):
        bwt, sa, r = sliced_emission(emu_factory, seqs, 4, 7, 4, ns)
        assert np.array_equal(bwt, ref["bwt"]) and np.array_equal(sa.astype(np.uint64), ref["sa"] & np.uint64(0xFFFFFFFF)) and r == ref["r"]
    sliced_samples(emu_factory, seqs, 4, 7, 4, 3, ref, sa=False)
    sliced_samples(emu_factory, seqs, 4, 7, 4, 2, ref, sa=True)


def test_sliced_samples_with_runless_slices_emu(emu_factory):
    """a text whose BWT is a few long runs: most slices hold no run start at all (only the last row's run end in the last one)"""
    seqs = [b"A" * 700 + b"C" * 500, b"A" * 900]
    ref = oracle_run(seqs, w=4, p=7, U=8)
    for ns in (4, 9):
        sliced_samples(emu_factory, seqs, 4, 7, 8, ns, ref, sa=False)
    sliced_samples(emu_factory, seqs, 4, 7, 8, 5, ref, sa=True)


@pytest.mark.gpu
def test_sliced_emission_gpu(gpu_ctx_factory):
    for seed, L, H, w, p, U in ((5, 200000, 4, 10, 100, 4), (6, 60000, 3, 4, 7, 8)):
        seqs = synth(seed, L, H)
        ref = oracle_run(seqs, w=w, p=p, U=U)
        for ns in (2, 3, 8):
            bwt, sa, r = sliced_emission(gpu_ctx_factory, seqs, w, p, U, ns)
            want = ref["sa"] & np.uint64(0xFFFFFFFF) if U == 4 else ref["sa"]
            assert np.array_equal(bwt, ref["bwt"]) and np.array_equal(sa.astype(np.uint64), want) and r == ref["r"]
        for ns, sa in ((2, False), (7, False), (3, True)):
            sliced_samples(gpu_ctx_factory, seqs, w, p, U, ns, ref, sa)


@pytest.mark.gpu
def test_sharded_merge_gpu(gpu_ctx_factory):
    seqs = synth(9, 300000, 6, (50000, 40000, 200000, 500))
    ref = oracle_run(seqs, w=10, p=100, U=4)
    for shards in ([[0, 1], [2, 3], [4, 5]], [[0], [1], [2], [3], [4], [5]], [[0, 1, 2, 3, 4], [5]]):
        assert compare(sharded_single_process(gpu_ctx_factory, seqs, shards, 10, 100, 4), ref, 4) == []
    seqs = synth(10, 60000, 4)
    ref = oracle_run(seqs, w=4, p=7, U=8)
    assert compare(sharded_single_process(gpu_ctx_factory, seqs, [[0], [1, 2], [3]], 4, 7, 8), ref, 8) == []
    assert compare(sharded_single_process(gpu_ctx_factory, seqs, [[0], [1, 2], [3]], 4, 7, 8, compact=True), ref, 8) == []


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["standalone", "loaded"])
def test_merge_of_standalone_shards_gpu(gpu_ctx_factory, mode):
    """merge_pfp's case on the card: shards parsed on their own (or loaded from saved .dict/.parse images) -- the seams are
    re-hashed inside pfp_merge_shards"""
    seqs = seam_trigger_seqs()
    for w, p, shards in ((4, 3, [[0], [1], [2], [3], [4], [5], [6]]), (6, 2, [[0, 1], [2], [3, 4, 5], [6]])):
        ref = oracle_run(seqs, w=w, p=p, U=8)
        assert compare(sharded_single_process(gpu_ctx_factory, seqs, shards, w, p, 8, mode=mode), ref, 8) == [], (mode, w, p)
    seqs = synth(9, 300000, 6, (50000, 40000, 200000, 500))
    ref = oracle_run(seqs, w=10, p=100, U=4)
    assert compare(sharded_single_process(gpu_ctx_factory, seqs, [[0, 1], [2, 3], [4, 5]], 10, 100, 4, mode=mode), ref, 4) == []


@pytest.mark.gpu
def test_merge_with_one_phrase_operands_gpu(gpu_ctx_factory):
    check_one_phrase_operands(gpu_ctx_factory, ("standalone", "loaded", "context"))


NCCL_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, os.path.join(sys.argv[1], "pfbwt-f_amd", "python"))
from pfp_testlib import compare, oracle_run
from test_sharded import synth
import pfbwt_hip, pfbwt_dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
seqs = synth(21, 200000, 5)
dev = torch.device("cuda", 0)
d = [torch.from_numpy(np.frombuffer(s, np.uint8).copy()).to(dev) for s in seqs]
ctx = pfbwt_hip.PfpContext(w=10, p=100, u64=True, sai=True, device=0)
feed = lambda c: [c.feed_device(t.data_ptr(), t.numel(), True) for t in d]
ref = oracle_run(seqs, w=10, p=100, U=8)
for sa, rssa in ((True, False), (False, True)):
    sz, b, begin, rows = pfbwt_dist.sharded_build(ctx, feed, 10, dev, sa=sa, rssa=rssa)
    o = ctx.bwt_get()
    res = {"n": sz.n, "m": sz.m, "dwords": sz.dwords, "dsize": sz.dsize, "r": int(b.r), "bwt": o["bwt"], "sa": o["sa"], "ssa": o["ssa"], "esa": o["esa"]}
    bad = compare(res, ref, 8, names=("bwt", "sa") if sa else ("bwt", "ssa", "esa"))
    assert bad == [] and begin == 0 and rows == sz.n + 1, bad
dist.barrier(); dist.destroy_process_group()
print("nccl world-1 ok")
'''


@pytest.mark.gpu
def test_sharded_build_rccl_world1(tmp_path):
    """pfbwt_dist.sharded_build through the RCCL backend on the one GPU of the test box (world size 1): the collective calls,
    the hand-over of the receive buffer from torch's stream to the engine's stream, merge of a single shard, slice 0 of 1"""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    pr = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert pr.returncode == 0 and "nccl world-1 ok" in pr.stdout, pr.stdout[-2000:] + pr.stderr[-3000:]


@pytest.mark.gpu
def test_sharded_c_api_gpu():
    """pfp_sharded_* on the MI355X: one rank through RCCL (ncclCommInitAll + ncclAllGather called by the library, world size 1),
    and the N-rank protocol rehearsed with three ranks on the one card (device-to-device exchange, N host threads)"""
    import pfbwt_hip
    assert pfbwt_hip.load_library().pfp_backend().decode() == "hip-gfx950"
    seqs = synth(9, 60000, 6)
    ref = oracle_run(seqs, w=10, p=100, U=8)
    for shards, devices in (([[0, 1, 2, 3, 4, 5]], None), ([[0, 1], [2], [3, 4, 5]], [0, 0, 0])):
        res = sharded_c_api(None, seqs, shards, 10, 100, 8, devices)
        assert compare(res, ref, 8, names=("bwt", "sa", "ssa", "esa")) == [] and res["r"] == ref["r"], devices
