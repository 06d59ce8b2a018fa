#!/usr/bin/env python
"""The tables of DESIGN.md that quote measurements, generated from the tracked files under profiles/<round>/ so that the
document cannot drift from them:  tools/design_tables.py [--write]  prints the blocks (or rewrites them in DESIGN.md between their
<!-- BEGIN name --> / <!-- END name --> markers).  tests/test_docs_tables.py fails when DESIGN.md and the files disagree."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = "r4"
P = os.path.join(ROOT, "profiles", RND)


def _traffic():
    out = {}
    f = os.path.join(P, "terms_hbm_traffic.txt")
    if not os.path.exists(f):
        return out
    for line in open(f):
        m = re.match(r"\s+(.*?) \[(.*?)\]\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+|-)\s+([\d.]+|-)", line)
        if m:
            out[m.group(1)] = {"tile": m.group(2), "total": float(m.group(5)), "x": float(m.group(6))}
    return out


def terms_table():
    t = json.load(open(os.path.join(P, "terms.json")))[0]
    tr = _traffic()
    rows = ["| stage kernel | tile | ms | algorithmic TB/s (÷ 8 TB/s) | HBM traffic (× algorithmic) | real TB/s |", "|---|---|---|---|---|---|"]
    for name, v in t.items():
        if not isinstance(v, dict):
            continue
        x = tr.get(name)
        ms = v["stage_ms"]
        rows.append("| %s | %s | %.3f | %.2f (%.2f) | %s | %s |" % (
            name, x["tile"] if x else "32x8", ms, v["GBs_algorithmic"] / 1e3, v["frac_of_8TBs"],
            "%.2f GB (%.2f)" % (x["total"], x["x"]) if x else "—", "%.2f" % (x["total"] / ms) if x else "—"))
    return "\n".join(rows)


def configs_table():
    rows = ["| what | figure | file |", "|---|---|---|"]
    b = json.load(open(os.path.join(P, "bench_default_run.json")))
    r = b["roofline"]
    rows.append("| headline, 512³ WENO5 advect + Eikonal RK3 (`bench.py --steps 20 --warmup 3`) | %.1f k Mcells/s, %.3f ms/step; stage kernel %.4f ms = %.0f GB/s algorithmic = **%.3f of HBM peak**; HBM traffic %s | `bench_default_run.json` |" % (
        b["value"] / 1e3, b["ms_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"],
        "%.2f GB per launch (%.2f × algorithmic)" % (r["traffic"] / 1e9, r["traffic"] / r["algorithmic_bytes_per_launch"]) if r.get("traffic") else "n/a"))
    c = b.get("cpu_baseline")
    if c:
        rows.append("| CPU beside it (same box, same run) | oracle %.2f Mcells/s on 1 core (%s); %.1f Mcells/s on %d cores | `bench_default_run.json` |" % (
            c["value"], c["sample"].split(" (")[0], c["all_cores"]["value"], c["all_cores"]["cores"]))
    f = os.path.join(P, "cpu_full.json")
    if os.path.exists(f):
        c = json.load(open(f))["cpu_baseline"]
        rows.append("| the full CPU sample of SURVEY.md §8d (`bench.py --cpu-full`) | %.2f Mcells/s on 1 core: %s | `cpu_full.json` |" % (c["value"], c["sample"]))
    c2 = json.load(open(os.path.join(P, "bench_config2.json")))["detail"]
    rows.append("| config 2, 2048² Zalesak | advection %.4f ms/step (stage %.4f ms, %.1f k Mcells/s), frozen-sign reinitialisation %.4f ms/step | `bench_config2.json`, `kernel_stats_config2.csv` |" % (
        c2["advect_ms_per_step"], c2["advect_stage_ms"], c2["advect_Mcells_s"] / 1e3, c2["reinit_ms_per_step"]))
    c3 = json.load(open(os.path.join(P, "bench_config3.json")))
    d3 = c3["detail"]
    rows.append("| config 3, 512³ NormalMotion + Curvature, ExtrapolationBC(2) | %.3f ms/step, %.1f k Mcells/s, stage %.4f ms = %.0f GB/s algorithmic = **%.3f of HBM peak** | `bench_config3.json`, `kernel_stats_config3.csv` |" % (
        d3["ms_per_step"], d3["Mcells_s"] / 1e3, d3["stage_ms"], d3["GBs_algorithmic"], d3["GBs_algorithmic"] / 8000.0))
    c5 = json.load(open(os.path.join(P, "bench_config5.json")))["detail"]
    rows.append("| config 5 on ONE device, 768³ float32 band, %.2f M active nodes | **%.3f ms/step** (float64 storage: %.3f) | `bench_config5.json`, `kernel_stats_config5.csv` |" % (
        c5["float32"]["active_nodes"] / 1e6, c5["float32"]["ms_per_step"], c5["float64"]["ms_per_step"]))
    f = os.path.join(P, "slab_overhead.json")
    if os.path.exists(f):
        s = json.load(open(f))
        w, o, pl = s["whole_grid_ms"], s["two_local_slabs_overlap_ms"], s["two_local_slabs_plain_ms"]
        rows.append("| slab decomposition on one device (two in-process ranks sharing the GPU, %s) | whole grid %.2f ms, two slabs %.2f ms with the boundary-first split (+%.1f %%), %.2f ms without (+%.1f %%); device-to-device copies stand in for xGMI | `slab_overhead.json` |" % (
            "×".join(str(k) for k in s["grid"]), w, o, 100 * (o / w - 1), pl, 100 * (pl / w - 1)))
    f = os.path.join(P, "bench_local8.json")
    if os.path.exists(f):
        b = json.load(open(f))
        rows.append("| the scaling run's code on ONE device (`bench.py --gpus 8 --transport local`: 8 rank threads, in-process transport, %s) | %.2f ms per step for the 8 slabs = %.3f ms per slab-step; self-check of the two stage orders: %s | `bench_local8.json` |" % (
            "×".join(str(k) for k in b["config"]["grid"]), b["ms_per_step"], b["ms_per_step"] / b["config"]["ranks"], b["config"]["halo_overlap"]))
    f = os.path.join(P, "reinit_bench.json")
    if os.path.exists(f):
        r = json.load(open(f))
        f5 = os.path.join(P, "reinit_bench_512.json")
        if os.path.exists(f5):
            r = r[:1] + [x for x in json.load(open(f5)) if x["band"]] + r[1:]
        rows.append("| `reinitialize!` (`tools/reinit_bench.py`) | " + "; ".join("%d³ %s, %.2f M nodes: %.2f ms" % (
            x["n"], "band" if x["band"] else "dense", x["active_nodes"] / 1e6, x["ms"]) for x in r) + " | `reinit_bench.json`, `reinit_bench_512.json`, `kernel_stats_reinit.csv` |")
    f = os.path.join(P, "bench_config5r.json")
    if os.path.exists(f):
        d = json.load(open(f))["detail"]
        rows.append("| config 5 with `reinitialize!` every %d steps (`bench.py --config 5r`) | %.3f ms per step, `reinitialize!` %.2f ms per call (%.2f M band nodes) | `bench_config5r.json` |" % (
            d["reinitialize_every"], d["ms_per_step"], d["reinitialize_ms"], d["active_nodes"] / 1e6))
    return "\n".join(rows)


BLOCKS = {"terms-table": terms_table, "configs-table": configs_table}


def render(text):
    for name, fn in BLOCKS.items():
        pat = re.compile(r"(<!-- BEGIN %s -->\n).*?(\n<!-- END %s -->)" % (name, name), re.S)
        if not pat.search(text):
            raise SystemExit(f"DESIGN.md has no block '{name}'")
        text = pat.sub(lambda m: m.group(1) + fn() + m.group(2), text)
    return text


if __name__ == "__main__":
    path = os.path.join(ROOT, "DESIGN.md")
    if "--write" in sys.argv:
        new = render(open(path).read())      # read (and render) before the file is opened for writing
        open(path, "w").write(new)
    else:
        for name, fn in BLOCKS.items():
            print(f"<!-- BEGIN {name} -->\n{fn()}\n<!-- END {name} -->\n")
