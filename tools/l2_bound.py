#!/usr/bin/env python
"""Developer tool (round 6, the round-5 verdict's item 2a): what the headline SpMM's REFERENCE STREAM costs at the fabric.

The gather of B is the product's dominant traffic: every nonzero (i, k) of A reads row k of B (512 B at N = 128 fp32), B
(537 MB) is 17 x the chip's L2 capacity, so what crosses the XCD <-> memory fabric is decided by which rows each XCD's
4 MiB L2 still holds.  This tool replays the references of the two schedules the library runs on the headline matrix

  row_owned  : chunks of 256 items (nonzeros + row ends), 2 column slices (XCD set s only touches columns s N/2 ..), chunk
               blocks dealt to the XCDs exactly as k_spmm maps them (csrc/spmm.hip:240-262);
  partitioned: rows of >= 64 entries split by kp_part(column) into 8 sub-matrices, sub-matrix q on XCD q only (whole rows of B),
               the other rows row-owned with 2 slices; 128-item chunks (the shipped defaults);

through ONE 4 MiB cache per XCD -- fully associative, `waves` chunks in flight per XCD interleaved round-robin 16 references
at a time (NG x U of the kernel) -- under LRU (what a real L2 approximates) and under Belady's optimal replacement (the fewest
misses ANY replacement policy could have on that order of references), and turns the misses into bytes and a floor in
milliseconds at the fabric rates measured on this machine (6.2 TB/s: what the product's own kernels reach on their mixed
traffic; 7.4 TB/s: the best gather probe).  Everything else the product must move regardless (A's entries once, C once, the
partial rows of the partitioned form twice) is added as compulsory bytes.

The matrix is the bench's generator (R-MAT scale 20, 32 edges / row, (.57,.19,.19,.05), seed 7) run on torch's CPU generator
-- the same distribution, not the same random numbers as the GPU run.  Usage: python tools/l2_bound.py [--scale 20] [--waves 256]
Prints one JSON object (also what bench.py embeds as roofline.formulation_floor)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HERE = os.path.join(ROOT, "tools", "l2_bound")


def rmat_cpu(scale, epr, seed):
    import torch
    n = 1 << scale
    ne = n * epr
    g = torch.Generator()
    g.manual_seed(seed)
    a, b, c = 0.57, 0.19, 0.19
    rows = torch.zeros(ne, dtype=torch.int64)
    cols = torch.zeros(ne, dtype=torch.int64)
    for _ in range(scale):
        r = torch.rand(ne, generator=g)
        rows = rows * 2 + (r >= a + b).long()
        cols = cols * 2 + (((r >= a) & (r < a + b)) | (r >= a + b + c)).long()
    key = torch.unique(rows * n + cols).numpy()
    r = key // n
    idx = (key % n).astype(np.int32)
    ptr = np.zeros(n + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(np.bincount(r, minlength=n))
    return ptr, idx, n


def interleave(chunk_of_ref, refs, n_chunks_total, my_chunks, waves, step=16):
    """Order in which ONE XCD issues its references: `my_chunks` (chunk ids, in dispatch order) run `waves` at a time; every turn each
    running chunk issues its next `step` references; a finished chunk's slot takes the next chunk."""
    # references of every chunk are contiguous in `refs` (sorted by chunk): starts / ends
    starts = np.searchsorted(chunk_of_ref, my_chunks, side="left")
    ends = np.searchsorted(chunk_of_ref, my_chunks, side="right")
    lens = ends - starts
    keep = lens > 0
    starts, lens = starts[keep], lens[keep]
    if starts.size == 0:
        return np.zeros(0, dtype=np.int32)
    # chunk j (in this XCD's order) occupies slot j % waves in "generation" j // waves; approximate the slot hand-over by
    # generations: all chunks of a generation run together, turn by turn (chunks are <= 256 references: <= 16 turns)
    out = []
    turns = int((lens.max() + step - 1) // step)
    ngen = (starts.size + waves - 1) // waves
    for gen in range(ngen):
        s = starts[gen * waves:(gen + 1) * waves]
        l = lens[gen * waves:(gen + 1) * waves]
        t_max = int((l.max() + step - 1) // step)
        for t in range(t_max):
            lo = t * step
            m = l > lo
            cnt = np.minimum(l[m] - lo, step)
            base = s[m] + lo
            # concatenate ranges base[i] .. base[i] + cnt[i]
            tot = int(cnt.sum())
            offs = np.repeat(base - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(tot)
            out.append(refs[offs])
    return np.concatenate(out).astype(np.int32)


def simulate(streams, universe, cap, exe):
    tot = {"refs": 0, "lru": 0, "opt": 0, "distinct": 0}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for k, s in enumerate(streams):
            if s.size == 0:
                continue
            path = os.path.join(tmp, "s%d.bin" % k)
            s.astype(np.int32).tofile(path)
            n, lru, opt, dis = (int(x) for x in subprocess.check_output([exe, path, str(universe), str(cap)]).split())
            tot["refs"] += n
            tot["lru"] += lru
            tot["opt"] += opt
            tot["distinct"] += dis
    return tot


def row_owned_streams(ptr, idx, n, chunk, slices, waves):
    """Per-XCD reference streams of the row-owned product (k_spmm's block mapping)."""
    nnz = idx.size
    row_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
    item = np.arange(nnz, dtype=np.int64) + row_of            # position of the nonzero in the (nonzeros + row ends) sequence
    chunk_of = (item // chunk).astype(np.int64)               # non-decreasing
    nchunks = int((nnz + n + chunk - 1) // chunk)
    per = 8 // slices
    nblocks = (nchunks + 3) // 4                              # 4 waves (chunks) per workgroup
    streams = []
    for x in range(8):
        cbs = np.arange(x % per, nblocks, per, dtype=np.int64)   # chunk blocks of this XCD, in dispatch order
        chunks = (cbs[:, None] * 4 + np.arange(4)[None, :]).ravel()
        chunks = chunks[chunks < nchunks]
        streams.append(interleave(chunk_of, idx, nchunks, chunks, waves))
    return streams


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ncols", type=int, default=128)
    ap.add_argument("--waves", type=int, default=256, help="chunks in flight per XCD (32 CUs x 8 waves)")
    ap.add_argument("--min-row", type=int, default=64)
    ap.add_argument("--achieved-ms", type=float, default=1.238)
    args = ap.parse_args()
    exe = os.path.join(HERE, "cache_sim")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".c"):
        subprocess.check_call(["gcc", "-O2", "-o", exe, exe + ".c"])
    ptr, idx, n = rmat_cpu(args.scale, 32, 7)
    nnz = int(idx.size)
    N = args.ncols
    l2 = 4 << 20
    out = {"matrix": "R-MAT 2^%d, %d nnz (CPU generator, same distribution as the bench's)" % (args.scale, nnz), "N": N,
           "l2_bytes_per_xcd": l2, "waves_in_flight_per_xcd": args.waves}
    compulsory = nnz * 8 + (n + 1) * 8 + n * N * 4            # A once, C once
    b_once = n * N * 4
    # ---- row-owned, 2 slices, 256-item chunks
    unit = N // 2 * 4
    st = row_owned_streams(ptr, idx, n, 256, 2, args.waves)
    r = simulate(st, n, l2 // unit, exe)
    out["row_owned"] = {"unit_bytes": unit, "references": r["refs"], "lru_miss_GB": round(r["lru"] * unit / 1e9, 3),
                        "opt_miss_GB": round(r["opt"] * unit / 1e9, 3), "compulsory_first_touch_GB": round(r["distinct"] * unit / 1e9, 3)}
    # ---- partitioned: long rows by kp_part(col) on XCD q (whole rows), short rows row-owned 2 slices; 128-item chunks
    lens = np.diff(ptr)
    is_long = lens >= args.min_row
    row_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    long_e = is_long[row_of]
    part = ((idx.astype(np.uint32).astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)
    part = (part & np.uint64(7)).astype(np.int64)
    n_long = int(is_long.sum())
    long_rank = np.cumsum(is_long) - 1                        # index of a long row among the long rows
    streams = []
    unit_l = N * 4
    for q in range(8):
        sel = long_e & (part == q)
        cols_q = idx[sel]
        rows_q = long_rank[row_of[sel]]                         # sub-row index inside partition q (rows in order)
        item = np.arange(cols_q.size, dtype=np.int64) + rows_q  # nonzeros + row ends of the sub-matrix
        chunk_of = item // 128
        nch = int((cols_q.size + n_long + 127) // 128)
        chunks = np.arange(nch, dtype=np.int64)                 # all of them on XCD q, in order
        streams.append(interleave(chunk_of, cols_q, nch, chunks, args.waves))
    rl = simulate(streams, n, l2 // unit_l, exe)
    # short rows: a matrix with the long rows emptied
    ptr_s = np.zeros(n + 1, dtype=np.int64)
    ptr_s[1:] = np.cumsum(np.where(is_long, 0, lens))
    idx_s = idx[~long_e]
    st = row_owned_streams(ptr_s, idx_s, n, 128, 2, args.waves)
    rs = simulate(st, n, l2 // unit, exe)
    partial = 2 * 8 * n_long * N * 4                            # written by the long-row kernel, read by the combine
    out["partitioned"] = {"long_rows": n_long, "nnz_in_long_rows": int(long_e.sum()),
                          "long": {"unit_bytes": unit_l, "references": rl["refs"], "lru_miss_GB": round(rl["lru"] * unit_l / 1e9, 3),
                                   "opt_miss_GB": round(rl["opt"] * unit_l / 1e9, 3)},
                          "short": {"unit_bytes": unit, "references": rs["refs"], "lru_miss_GB": round(rs["lru"] * unit / 1e9, 3),
                                    "opt_miss_GB": round(rs["opt"] * unit / 1e9, 3)},
                          "partial_rows_GB": round(partial / 1e9, 3)}
    for name, b_lru, b_opt, extra in (("row_owned", out["row_owned"]["lru_miss_GB"], out["row_owned"]["opt_miss_GB"], 0.0),
                                      ("partitioned", out["partitioned"]["long"]["lru_miss_GB"] + out["partitioned"]["short"]["lru_miss_GB"],
                                       out["partitioned"]["long"]["opt_miss_GB"] + out["partitioned"]["short"]["opt_miss_GB"], partial / 1e9)):
        tot_lru = b_lru + compulsory / 1e9 + extra
        tot_opt = b_opt + compulsory / 1e9 + extra
        out[name]["fabric_GB_lru"] = round(tot_lru, 3)
        out[name]["fabric_GB_opt"] = round(tot_opt, 3)
        out[name]["floor_ms_at_6.2TBps"] = {"lru": round(tot_lru / 6.2, 4), "opt": round(tot_opt / 6.2, 4)}
        out[name]["floor_ms_at_7.4TBps"] = {"lru": round(tot_lru / 7.4, 4), "opt": round(tot_opt / 7.4, 4)}
    out["compulsory_GB"] = round(compulsory / 1e9, 3)
    out["b_once_GB"] = round(b_once / 1e9, 3)
    out["achieved_ms"] = args.achieved_ms
    out["achieved_over_opt_floor_at_6.2TBps"] = round(args.achieved_ms / out["partitioned"]["floor_ms_at_6.2TBps"]["opt"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
