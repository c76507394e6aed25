"""Per-kernel roofline of ONE NDT1-CTC train step (bench.py's workload) from rocprofv3 output -> profiles/rNN_roofline.json.

Inputs (all produced on the GPU box by ONE gpurun call, see profiles/README.md for the exact commands):
  --trace  DB    rocprofv3 --kernel-trace --stats  (SQLite, the `kernels` view)       of `python3 bench.py ...`
  --log    FILE  NBCI_GEMM_LOG of the SAME run: one line per GEMM launch (M N K ...), joined with the trace in launch order
  --pmc    DB... rocprofv3 --pmc passes (own runs, each with its own --pmc-log): FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES ...
  --pmc-log FILE... the NBCI_GEMM_LOG of each --pmc run (same order)

For every kernel class of a mid-run step: launches, average duration, ALGORITHMIC work (GEMMs: 2 M N K from the launch log;
the memory-bound kernels: operands read once + results written once, from the workload's shapes), achieved rate, the bound
(MFMA or HBM) and the fraction of its peak (MI355X_MICROARCH.md: 2.5 PFLOP/s dense bf16 MFMA, 157.3 TFLOP/s f32 MFMA, 8 TB/s
HBM3E); from the counter passes: HBM-side bytes per launch (FETCH_SIZE doubled per the gfx950 correction, KiB -> bytes;
WRITE_SIZE KiB -> bytes) and the matrix-pipe busy fraction SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs).
"""
import argparse
import collections
import json
import re
import sqlite3

PEAK_BF16, PEAK_F32, PEAK_HBM = 2500.0, 157.3, 8000.0   # TFLOP/s, TFLOP/s, GB/s


def short(name):
    n = name.replace("nbci::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    m = re.match(r"_ZN4nbci(\d+)([a-z_0-9]+)", n)
    if m:
        n = m.group(2)[:int(m.group(1))]
    return n[:70]


def is_gemm(name):
    return "gemm_" in name and "rocclr" not in name


def read_log(path):
    out = []
    for line in open(path):
        p = line.split()
        if p[0] == "S":
            M, N, K, batch, kind = (int(x) for x in p[1:6])
            out.append({"flops": 2.0 * M * N * K * batch, "shape": f"{M}x{N}x{K}" + (f" x{batch}" if batch > 1 else ""), "kind": kind})
        else:
            n, kind = int(p[1]), int(p[2])
            v = [int(x) for x in p[3:]]
            fl = sum(2.0 * v[3 * i] * v[3 * i + 1] * v[3 * i + 2] for i in range(n))
            out.append({"flops": fl, "shape": "group[" + ", ".join(f"{v[3*i]}x{v[3*i+1]}x{v[3*i+2]}" for i in range(n)) + "]", "kind": kind})
    return out


def load_kernels(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start").fetchall()
    return [dict(name=r[0], start=r[1], end=r[2], grid=r[3] // max(1, r[4]), vgpr=r[5], agpr=r[6], lds=r[7]) for r in rows]


def join_log(kernels, log):
    g = [k for k in kernels if is_gemm(k["name"])]
    if len(g) != len(log):
        raise SystemExit(f"trace has {len(g)} GEMM launches, the log {len(log)}: not the same run")
    for k, l in zip(g, log):
        k.update(l)


def mid_step(kernels):
    idx = [i for i, k in enumerate(kernels) if "per_kernel" in k["name"]]
    mid = len(idx) // 2
    return kernels[idx[mid - 1] + 1: idx[mid] + 1]


def memory_model(B, T, N, H, L, V, Tp, params):
    """algorithmic bytes per launch of the HBM-bound kernels (operands once in, results once out) for the bench workload"""
    M = B * Tp
    return {
        "ln_fwd_kernel": ("LayerNorm forward: x f32 in, h bf16 out", M * H * 4 + M * H * 2),
        "ln_bwd_kernel": ("LayerNorm backward (+ fused cast / bias sums): dy bf16, x f32, dx f32 in/out, bf16 cast out", M * H * (2 + 4 + 4 + 4 + 2)),
        "adamw_kernel": ("AdamW over the flat buffer: w, g, m, v in; w, m, v, bf16 shadow out", params * 30),
        "smooth_noise_kernel": ("smoothing + noise: spikes f32 in, bf16 out", B * T * N * 4 + B * T * N * 2),
        "colsum_kernel": ("bias-gradient column sums", None),
        "posgrad_kernel": ("position-table gradient", M * H * 4),
        "FillFunctor": ("zero_grad fill", None),
    }


def attention_flops(B, heads, Tp, hd):
    f = 4.0 * Tp * Tp * hd * B * heads
    return {"attn_fwd_kernel": f, "attn_bwd_dq_kernel": 1.5 * f, "attn_bwd_dkv_kernel": 1.0 * f}   # bwd: dq = S,dP,dQ ; dkv = dV,dK (2.5x in all)


def pmc_table(db, log):
    """{(short name, grid, shape): {counter: mean value per launch}}"""
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, grid_size_x, workgroup_size_x, counter_name, value, dispatch_id from counters_collection order by dispatch_id").fetchall()
    disp = collections.OrderedDict()
    for name, gx, wx, cn, v, did in rows:
        d = disp.setdefault(did, {"name": name, "grid": gx // max(1, wx), "c": {}})
        d["c"][cn] = d["c"].get(cn, 0.0) + float(v)
    ds = list(disp.values())
    g = [d for d in ds if is_gemm(d["name"])]
    if log is not None:
        if len(g) != len(log):
            raise SystemExit(f"pmc run has {len(g)} GEMM launches, its log {len(log)}")
        for d, l in zip(g, log):
            d["shape"] = l["shape"]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ds:
        key = (short(d["name"]), d["grid"], d.get("shape"))
        for cn, v in d["c"].items():
            acc[key][cn].append(v)
    return {k: {cn: sum(v) / len(v) for cn, v in cs.items()} for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", required=True)
    ap.add_argument("--log", required=True)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--pmc-log", nargs="*", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--bins", type=int, default=600)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    B, T, N, H, L, V, heads = a.batch, a.bins, a.channels, 1024, 5, 41, 8
    Tp = 1 + (T - 32) // 4
    params = 41056553
    ks = load_kernels(a.trace)
    join_log(ks, read_log(a.log))
    step = mid_step(ks)
    span = (step[-1]["end"] - step[0]["start"]) / 1e3
    mem = memory_model(B, T, N, H, L, V, Tp, params)
    attn = attention_flops(B, heads, Tp, H // heads)
    pmc = {}
    for i, db in enumerate(a.pmc):
        lg = read_log(a.pmc_log[i]) if i < len(a.pmc_log) else None
        for k, v in pmc_table(db, lg).items():
            pmc.setdefault(k, {}).update(v)
    classes = collections.OrderedDict()
    for k in step:
        key = (short(k["name"]), k["grid"], k.get("shape"))
        c = classes.setdefault(key, {"us": [], "flops": k.get("flops"), "vgpr": k["vgpr"], "agpr": k["agpr"], "lds": k["lds"], "kind": k.get("kind")})
        c["us"].append((k["end"] - k["start"]) / 1e3)
    rows, tot = [], 0.0
    for (nm, grid, shape), c in classes.items():
        n, avg = len(c["us"]), sum(c["us"]) / len(c["us"])
        tot += sum(c["us"])
        r = {"kernel": nm, "workgroups": grid, "launches_per_step": n, "avg_us": round(avg, 2), "us_per_step": round(sum(c["us"]), 1),
             "vgpr": c["vgpr"], "agpr": c["agpr"], "lds_bytes": c["lds"]}
        if shape:
            r["problem"] = shape
        fl = c["flops"] if c["flops"] else next((f for kn, f in attn.items() if kn in nm), None)
        if fl:
            peak = PEAK_F32 if (c["kind"] is not None and c["kind"] < 4) else PEAK_BF16
            r.update(bound="mfma", algorithmic_gflop=round(fl / 1e9, 2), achieved_tflops=round(fl / avg / 1e6, 1), peak_tflops=peak,
                     frac_of_peak=round(fl / avg / 1e6 / peak, 4))
        else:
            mm = next((v for kn, v in mem.items() if kn in nm), None)
            if mm and mm[1]:
                r.update(bound="hbm", what=mm[0], algorithmic_mbytes=round(mm[1] / 1e6, 1), achieved_gbs=round(mm[1] / avg / 1e3, 0),
                         peak_gbs=PEAK_HBM, frac_of_peak=round(mm[1] / avg / 1e3 / PEAK_HBM, 4))
            else:
                r.update(bound="latency")
        pc = pmc.get((nm, grid, shape)) or pmc.get((nm, grid, None))
        if pc:
            if "FETCH_SIZE" in pc:
                r["hbm_fetch_mbytes"] = round(2.0 * 1024 * pc["FETCH_SIZE"] / 1e6, 1)
            if "WRITE_SIZE" in pc:
                r["hbm_write_mbytes"] = round(1024 * pc["WRITE_SIZE"] / 1e6, 1)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in pc and pc.get("GRBM_GUI_ACTIVE"):
                r["mfma_busy_frac"] = round(pc["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * pc["GRBM_GUI_ACTIVE"] / 8.0), 4)
            for cn in ("SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                if cn in pc:
                    r.setdefault("counters", {})[cn] = round(pc[cn])
        rows.append(r)
    rows.sort(key=lambda r: -r["us_per_step"])
    out = {"workload": f"bench.py: NDT1-CTC train step, B={B}, {N} ch x {T} bins -> {Tp} tokens, bf16", "note": a.note,
           "step_span_us": round(span, 1), "sum_of_kernel_us": round(tot, 1),
           "peaks": {"bf16_mfma_tflops": PEAK_BF16, "f32_mfma_tflops": PEAK_F32, "hbm_gbs": PEAK_HBM},
           "corrections": "FETCH_SIZE x 2 x 1024 B (gfx950 tallies 128-B requests at 64 B; KiB units), WRITE_SIZE x 1024 B; mfma_busy_frac = "
                          "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)",
           "kernels": rows}
    json.dump(out, open(a.out, "w"), indent=1)
    print(f"step span {span:.1f} us, kernel sum {tot:.1f} us, {len(rows)} kernel classes -> {a.out}")
    for r in rows[:14]:
        extra = f"{r.get('achieved_tflops', r.get('achieved_gbs', ''))} {'TF/s' if r.get('bound') == 'mfma' else ('GB/s' if r.get('bound') == 'hbm' else '')}"
        print(f"  {r['kernel'][:44]:44s} {r.get('problem', ''):34s} n={r['launches_per_step']:2d} avg {r['avg_us']:7.1f} us  {extra:14s} frac {r.get('frac_of_peak', '')}"
              f"  mfma_busy {r.get('mfma_busy_frac', '')}")


if __name__ == "__main__":
    main()
