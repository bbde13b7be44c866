"""Host-side checks of the walk-plan selection (csrc/select3.hip) without a GPU: the index builder
(``lpformer_amd.graph.build_walk_index``, plain torch, runs on CPU tensors) and a line-by-line Python emulation of the
plan + run kernels' per-slot logic over those indexes, compared bit for bit with the oracle's ``select_nodes``
(src/models/link_transformer.py:214-319, 434-481).  What this pins is the ALGEBRA of the path -- every selected set
written as an intersection evaluated from its shorter side through the hashed union index -- on the cases where it could
go wrong: a == b, adjacent endpoints, duplicates, isolated nodes, theta_1 <= 0 (absent PPR entries pass), theta_cn > 0,
theta_n < theta_1, PPR values within a few ulp of the thresholds, mask modes "all", "1-hop" and "cn"."""
import numpy as np
import pytest
import torch

from lpformer_amd import graph
from oracle import lpformer_oracle as O

F32 = np.float32
HASH_MUL = 2654435761
K_FULL, K_A1, K_PX, K_T0 = 0, 1, 2, 3


def _dev(c: graph.CSR) -> graph.DeviceCSR:
    return graph.DeviceCSR(torch.from_numpy(c.rowptr), torch.from_numpy(c.col),
                           None if c.val is None else torch.from_numpy(c.val), c.n, c)


def _bloom_hash(v: int) -> int:
    """select3.hip::s3_bloom_hash / lpformer_amd.graph.bloom_hash, on a Python int."""
    h = (v * graph.BLOOM_MUL1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * graph.BLOOM_MUL2) & 0xFFFFFFFF
    return h ^ (h >> 13)


def _mini_pass(mini_row, x: int) -> bool:
    """The run kernel's test of candidate x against a node's 1,024-bit mini filter (staged in LDS there)."""
    h = _bloom_hash(int(x) ^ graph.MINI_SALT)
    w = int(mini_row[h >> 27]) & 0xFFFFFFFF
    return bool((w >> (h & 31)) & (w >> ((h >> 5) & 31)) & 1)


def _rt1(p):
    return F32(F32(F32(p) + F32(1)) - F32(1))


def _rt2(p):
    return F32(F32(0.5) * F32(F32(F32(p) * F32(2) + F32(2)) - F32(2)))


def emulate_select3(wi: graph.WalkIndex, batch, thresholds, mode, slot_order=None):
    """The plan kernel's walk choice and the run kernel's slot typing (select3.hip / walk_common.h), one slot at a time.
    ``slot_order``: a list that receives every kept slot as (pair, code, node, pa, pb, from_N(b)) in candidate-slot
    order -- the order lpf_select4 compacts in."""
    th_cn, th_1, th_n = (F32(t) for t in thresholds)
    rec = wi.rec.numpy()
    r64 = wi.rec.view(torch.int64).numpy()
    cvs = {"adj": wi.adj_cv.numpy(), "a1": wi.a1_cv.numpy(), "px": wi.px_cv.numpy(),
           "t0": None if wi.t0_cv is None else wi.t0_cv.numpy()}
    ucv = wi.u.cv.numpy()
    mode_cn = mode == "cn"
    out = {1: [], 2: [], 3: []}

    def node(i):
        return dict(id=i, adj0=r64[i, 0], a10=r64[i, 1], px0=r64[i, 2], t00=r64[i, 3], u0=r64[i, 4], deg=rec[i, 10],
                    n_a1=rec[i, 11], n_px=rec[i, 12], n_t0=rec[i, 13], unb=rec[i, 14])

    mini = wi.mini.numpy()

    def lookup(look, x):
        u0, unb = look["u0"], look["unb"]
        if unb <= 0:
            return False, False, F32(0)
        # the looked-up endpoint's mini filter: a candidate that fails it is not in the row, its bucket is never read
        if not _mini_pass(mini[look["id"]], x):
            return False, False, F32(0)
        b = ((int(x) * HASH_MUL % 2**32) * int(unb)) >> 32
        blk = ucv[u0 + 8 * b: u0 + 8 * b + 8]
        for c, v in blk:
            if c == x:
                bits = int(v) & 0xFFFFFFFF
                return True, bool(bits >> 31), np.array([bits & 0x7FFFFFFF], np.uint32).view(F32)[0]
        return False, False, F32(0)

    for k in range(batch.shape[1]):
        a, b = int(batch[0, k]), int(batch[1, k])
        r = [node(a), node(b)]
        s = 0 if r[0]["deg"] <= r[1]["deg"] else 1
        walks = []
        for e in (0, 1):
            o = 1 - e
            if e == s:
                walks.append((K_FULL, "adj", r[e]["adj0"], r[e]["deg"], r[o], e == 0, e == 1))
            elif mode_cn:
                continue
            elif wi.use_px and r[o]["n_px"] < r[e]["n_a1"]:
                walks.append((K_PX, "px", r[o]["px0"], r[o]["n_px"], r[e], o == 0, e == 1))
            else:
                walks.append((K_A1, "a1", r[e]["a10"], r[e]["n_a1"], r[o], e == 0, e == 1))
        if cvs["t0"] is not None:
            e = 0 if r[0]["n_t0"] <= r[1]["n_t0"] else 1
            walks.append((K_T0, "t0", r[e]["t00"], r[e]["n_t0"], r[1 - e], e == 0, False))
        for kind, arr, src0, ln, look, src_a, side_b in walks:
            for j in range(int(ln)):
                x, bits = cvs[arr][src0 + j]
                ws = np.array([bits], np.int32).view(F32)[0]
                found, adj, lv = lookup(look, x)
                cn = kind == K_FULL and adj
                hop = (kind in (K_FULL, K_A1) and not adj) or (kind == K_PX and adj)
                far = kind == K_T0 and found and not adj
                two = cn and not mode_cn
                rs, rl = (_rt2(ws), _rt2(lv)) if two else (_rt1(ws), _rt1(lv))
                c = 0
                if cn:
                    c = 1 if (rs >= th_cn and rl >= th_cn) else 0
                elif hop:
                    c = 2 if (not mode_cn and rs >= th_1 and rl >= th_1) else 0
                elif far:
                    c = 3 if (ws > 0 and lv > 0 and rs >= th_n and rl >= th_n) else 0
                if c:
                    out[c].append((k, int(x), rs if src_a else rl, rl if src_a else rs, side_b))
                    if slot_order is not None:
                        slot_order.append((k, c, int(x), rs if src_a else rl, rl if src_a else rs, side_b))
    res = {}
    for c, tag in ((1, "cn"), (2, "onehop"), (3, "non1hop")):
        ent = out[c]
        if tag == "onehop":  # the export kernel merges the N(a) run and the N(b) run by node id
            ent = sorted(ent, key=lambda t: (t[0], t[1]))
        ix = np.array([[t[0] for t in ent], [t[1] for t in ent]], np.int64).reshape(2, -1)
        res[tag] = (ix, np.array([t[2] for t in ent], F32), np.array([t[3] for t in ent], F32))
    return res


def emulate_select4(wi: graph.WalkIndex, batch, thresholds, mode, blocks_per_wg=2):
    """What lpf_select4 (csrc/select4.hip) leaves, restated on the host: the kept slots of a workgroup's blocks of 64
    pairs compacted in candidate-slot order (pair-major: a pair's slots are contiguous), the type and the from-N(b) flag
    in the record's pair word, a table entry {first entry, n_cn, n_1hop, n_non1hop} per pair, {entries, pairs with
    entries} per block of 64 pairs.  Where a workgroup's run lands is an atomic add on the device; here the runs
    simply follow each other, rounded up to 8 entries like there."""
    kept = []
    emulate_select3(wi, batch, thresholds, mode, slot_order=kept)
    bs = batch.shape[1]
    per_pair = [[] for _ in range(bs)]
    for k, c, x, pa, pb, side_b in kept:      # (slot order inside every pair; pairs in batch order)
        per_pair[k].append((c, x, pa, pb, side_b))
    entries, tab = [], np.zeros((bs, 4), np.int64)
    blk = np.zeros(((bs + 63) // 64, 2), np.int64)
    wg_pairs = 64 * blocks_per_wg
    for w0 in range(0, bs, wg_pairs):
        for k in range(w0, min(w0 + wg_pairs, bs)):
            tab[k] = [len(entries)] + [sum(1 for e in per_pair[k] if e[0] == t) for t in (1, 2, 3)]
            for c, x, pa, pb, side_b in per_pair[k]:
                word = (k | (c << 29) | ((1 << 31) if (c == 2 and side_b) else 0)) & 0xFFFFFFFF
                entries.append((word, x, pa, pb))
            blk[k // 64, 0] += len(per_pair[k])
            blk[k // 64, 1] += 1 if per_pair[k] else 0
        entries.extend([(0, 0, F32(0), F32(0))] * (-len(entries) % 8))     # (the next run starts on a 128-byte line)
    return entries, tab, blk


def _case(seed, n, m, thresholds, jitter=True):
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, n, size=(2, m))
    ei = ei[:, ei[0] != ei[1]]
    ei = np.concatenate([ei, ei[::-1]], axis=1)
    adj = graph.mask_csr(ei, n)
    # a synthetic "PPR" matrix: diagonal 0.15+, neighbours and random far nodes, values clustered around the thresholds
    rows, cols, vals = [], [], []
    ths = [t for t in thresholds if 0 < t < 1] or [1e-2]
    for i in range(n):
        if i % 17 == 3:
            continue  # a row with nothing stored at all
        nb = adj.col[adj.rowptr[i]:adj.rowptr[i + 1]]
        far = rng.integers(0, n, size=rng.integers(0, 12))
        cs = np.unique(np.concatenate([[i], nb[rng.random(nb.size) < 0.8], far]))
        v = rng.choice(ths, size=cs.size).astype(F32)
        if jitter:  # +-6 ulp of fl32(1 + theta) => decides the round-tripped comparison
            v = (v + (rng.integers(-6, 7, size=cs.size) * F32(2.0 ** -23)).astype(F32)).astype(F32)
        v[rng.random(cs.size) < 0.3] *= F32(3.7)
        v = np.abs(v)
        v[cs == i] = F32(0.15) + rng.random(1).astype(F32)[0] * F32(0.2)
        rows.append(np.full(cs.size, i)); cols.append(cs); vals.append(v)
    ppr = graph.csr_from_coo(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), n)
    bs = 160
    batch = rng.integers(0, n, size=(2, bs))
    batch[:, :8] = batch[0, :8]                       # a == b
    k = min(20, ei.shape[1])
    batch[:, 8:8 + k] = ei[:, rng.integers(0, ei.shape[1], size=k)]   # adjacent endpoints
    batch[:, 30:34] = batch[:, 8:12]                  # duplicates
    iso = np.setdiff1d(np.arange(n), np.unique(ei))[:4]
    batch[0, 40:40 + iso.size] = iso                  # isolated endpoints
    return adj, ppr, batch


CASES = [
    ("all", (0.0, 1e-3, 1e-2)), ("all", (1e-3, 1e-2, 1e-2)), ("all", (0.0, 1e-2, 1e-3)),   # theta_n < theta_1
    ("all", (0.0, 0.0, 1e-2)), ("all", (2e-3, -1.0, 0.0)),                                    # theta_1 <= 0
    ("1-hop", (0.0, 1e-2, 1.0)), ("1-hop", (1e-2, 0.0, 1.0)), ("cn", (0.0, 1.0, 1.0)), ("cn", (1e-2, 1.0, 1.0)),
]


@pytest.mark.parametrize("mode,thresholds", CASES)
@pytest.mark.parametrize("seed", [0, 1])
def test_walk_plan_matches_oracle(mode, thresholds, seed):
    n = 220
    adj, ppr, batch = _case(seed, n, 900 if seed else 500, thresholds)
    wi = graph.build_walk_index(_dev(adj), _dev(ppr), thresholds[1], thresholds[2], want_t0=(mode == "all"))
    got = emulate_select3(wi, batch, thresholds, mode)
    ref = O.select_nodes(batch, (adj.rowptr, adj.col.astype(np.int64)),
                         (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), thresholds, n=n)
    for tag in ("cn", "onehop", "non1hop"):
        if tag not in ref:
            assert got[tag][0].shape[1] == 0
            continue
        assert np.array_equal(got[tag][0], ref[tag][0]), (mode, tag)
        assert np.array_equal(got[tag][1].view(np.uint32), ref[tag][1].view(np.uint32)), (mode, tag, "pa")
        assert np.array_equal(got[tag][2].view(np.uint32), ref[tag][2].view(np.uint32)), (mode, tag, "pb")
    assert sum(ref[t][0].shape[1] for t in ref) > 10  # the case selects something


@pytest.mark.parametrize("mode,thresholds", [CASES[0], CASES[3], CASES[5], CASES[7]])
def test_pair_major_layout_matches_oracle(mode, thresholds):
    """The pair-major layout of lpf_select4, restated on the host (``emulate_select4``): every pair's entries
    contiguous from its table entry, per-type counts and block counts consistent, the 1-hop entries of N(a) in front of
    those of N(b) -- and, regrouped by type and sorted by (pair, node), exactly the oracle's sets with the oracle's
    values.  (The device kernel is checked against the same oracle in tests/test_gpu_select4.py.)"""
    n = 220
    adj, ppr, batch = _case(1, n, 900, thresholds)
    wi = graph.build_walk_index(_dev(adj), _dev(ppr), thresholds[1], thresholds[2], want_t0=(mode == "all"))
    entries, tab, blk = emulate_select4(wi, batch, thresholds, mode)
    ref = O.select_nodes(batch, (adj.rowptr, adj.col.astype(np.int64)),
                         (ppr.rowptr, ppr.col.astype(np.int64), ppr.val), thresholds, n=n)
    bs = batch.shape[1]
    got = {t: [] for t in (1, 2, 3)}
    for k in range(bs):
        s0, cnt = tab[k, 0], tab[k, 1:].sum()
        seen_b = False
        for word, x, pa, pb in entries[s0:s0 + cnt]:
            assert (word & 0x1FFFFFFF) == k                       # the pair's own entries, nothing else
            t = (word >> 29) & 3
            got[t].append((k, x, pa, pb))
            if t == 2:                                            # N(a)'s one-hop nodes before N(b)'s
                assert not (seen_b and not (word >> 31))
                seen_b = seen_b or bool(word >> 31)
        for t in (1, 2, 3):
            assert sum(1 for e in entries[s0:s0 + cnt] if ((e[0] >> 29) & 3) == t) == tab[k, t]
    for b in range(blk.shape[0]):
        assert blk[b, 0] == tab[64 * b:64 * b + 64, 1:].sum() and blk[b, 1] == (tab[64 * b:64 * b + 64, 1:].sum(1) > 0).sum()
    for t, tag in ((1, "cn"), (2, "onehop"), (3, "non1hop")):
        ent = sorted(got[t], key=lambda e: (e[0], e[1]))
        if tag not in ref:
            assert not ent
            continue
        assert np.array_equal(np.array([[e[0] for e in ent], [e[1] for e in ent]], np.int64).reshape(2, -1), ref[tag][0])
        assert np.array_equal(np.array([e[2] for e in ent], F32).view(np.uint32), ref[tag][1].view(np.uint32))
        assert np.array_equal(np.array([e[3] for e in ent], F32).view(np.uint32), ref[tag][2].view(np.uint32))


def test_walk_index_layout():
    """Every adjacency entry and every px entry is found in its hashed bucket with the right flag and value; a1 / px / t0
    rows hold exactly the entries their definitions name."""
    n, th = 150, (0.0, 1e-2, 2e-2)
    adj, ppr, _ = _case(3, n, 600, th)
    wi = graph.build_walk_index(_dev(adj), _dev(ppr), th[1], th[2], want_t0=True)
    rec, r64, ucv = wi.rec.numpy(), wi.rec.view(torch.int64).numpy(), wi.u.cv.numpy()
    selfp = graph.self_ppr(adj, ppr)
    assert np.array_equal(wi.adj_cv.numpy()[:, 0], adj.col) and np.array_equal(wi.adj_cv.numpy()[:, 1].view(F32), selfp)
    one = F32(1)
    mini, fp_total = wi.mini.numpy(), [0, 0]
    assert mini.shape == (n, graph.MINI_WORDS)
    for i in range(n):
        nb = adj.col[adj.rowptr[i]:adj.rowptr[i + 1]]
        sp = selfp[adj.rowptr[i]:adj.rowptr[i + 1]]
        strong = ((sp + one) - one) >= F32(th[1])
        a1 = wi.a1_cv.numpy()[r64[i, 1]: r64[i, 1] + rec[i, 11]]
        assert np.array_equal(a1[:, 0], nb[strong]) and np.array_equal(a1[:, 1].view(F32), sp[strong])
        pc, pv = ppr.col[ppr.rowptr[i]:ppr.rowptr[i + 1]], ppr.val[ppr.rowptr[i]:ppr.rowptr[i + 1]]
        keep = (((pv + one) - one) >= F32(min(th[1], th[2]))) & ~np.isin(pc, nb)
        px = wi.px_cv.numpy()[r64[i, 2]: r64[i, 2] + rec[i, 12]]
        assert np.array_equal(px[:, 0], pc[keep]) and np.array_equal(px[:, 1].view(F32), pv[keep])
        far = keep & (pv > 0) & (((pv + one) - one) >= F32(th[2]))
        t0 = wi.t0_cv.numpy()[r64[i, 3]: r64[i, 3] + rec[i, 13]]
        assert np.array_equal(t0[:, 0], pc[far])
        want = {int(c): (True, v) for c, v in zip(nb, sp)}
        want.update({int(c): (False, v) for c, v in zip(pc[keep], pv[keep])})
        row = ucv[r64[i, 4]: r64[i, 4] + 8 * rec[i, 14]]
        live = row[row[:, 0] != 2**31 - 1]
        assert live.shape[0] == len(want)
        for c, bits in live:
            b = ((int(c) * HASH_MUL % 2**32) * int(rec[i, 14])) >> 32
            assert (row[8 * b: 8 * b + 8, 0] == c).sum() == 1          # in the bucket its hash names
            flag, v = want[int(c)]
            assert bool((int(bits) >> 31) & 1) == flag
            assert np.array([int(bits) & 0x7FFFFFFF], np.uint32).view(F32)[0] == v
            assert _mini_pass(mini[i], int(c))                          # the mini filter has no false negatives
        absent = [c for c in range(n) if c not in want]
        fp_total[0] += sum(_mini_pass(mini[i], c) for c in absent)
        fp_total[1] += len(absent)
    assert fp_total[0] <= 0.05 * fp_total[1], fp_total                  # rows of a few dozen keys in 1,024 bits
