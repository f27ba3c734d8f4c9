"""Roofline records with counter traffic for the launches of tools/pmc_family_bl.sh: writes gpurun_out/<tag>_pmc_family.json (copied to
profiles/, read by bench.py) -- per launch: kernel, measured us under the profiler, FETCH_SIZE (doubled: gfx950 counts the 128-byte
requests of 16-byte-per-lane loads at 64 B, MI355X_MICROARCH.md), WRITE_SIZE, L2 hit rate, algorithmic bytes, FLOPs, mixed roof.
Usage: pmc_family_summary.py <tag> <commit>"""
import csv, json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, commit = sys.argv[1], sys.argv[2]
O = os.path.join(R, "gpurun_out")
PEAK, HBM = 2.5e15, 8.0e12


def rows(layer, which):
    return list(csv.DictReader(open(os.path.join(O, f"{tag}_pmc_{layer}_{which}.csv"))))


def pick(layer, which, match, blocks=None):
    c = [r for r in rows(layer, which) if match in r["kernel"] and (blocks is None or int(r["blocks"]) == blocks)]
    # the layer bench launches forward, input gradient and weight gradient: the forward is the shortest launch of its kernel (the
    # phases-as-rows input gradient of MelGAN L4 runs the SAME tap4 instantiation over twice the rows), the others match one launch each
    c.sort(key=lambda r: float(r["us"]) if "tap4" in match else -float(r["us"]))
    return c[0]


# (record name, layer, kernel match, blocks or None, description, flops, algorithmic bytes) at 64 rows (forward / weight gradient)
rows64 = 64
spec = [
    ("melgan_l4_fwd", "melgan.4", "tap4_kernelILi2ELi2ELi4ELi2ELi1E", 256, "MelGAN L4 forward (1024->1024 k41 s4 g4, 500->125), 64 rows, bf16 bundles",
     2.0 * rows64 * 1024 * 256 * 41 * 125, rows64 * (2 * 1024 * 500 + 4 * 1024 * 125) + 2 * 1024 * 256 * 41),
    ("melgan_l4_dw", "melgan.4", "bl_dw_kernel", None, "MelGAN L4 weight gradient, 64 rows [fake | real] x [enhanced | reference]",
     2.0 * rows64 * 1024 * 256 * 41 * 125, rows64 * 2 * (1024 * 500 + 1024 * 125) + 4 * 1024 * 256 * 41),
    ("pqmf_l4_fwd", "pqmf0.4", "ELi2ELi2ELb1E", None, "PQMF-band discriminator L4 forward (192->384 k7 s2 g4, 1000->497), 64 rows, hi + lo operands",
     2.0 * 3 * rows64 * 384 * 48 * 7 * 497, rows64 * (4 * 192 * 1000 + 4 * 384 * 497) + 4 * 384 * 48 * 7),
]
out = []
for name, layer, match, blocks, desc, flops, alg in spec:
    f, w, l2 = pick(layer, "fetch", match, blocks), pick(layer, "write", match, blocks), pick(layer, "l2", match, blocks)
    fetch, write = 2 * float(f["FETCH_SIZE"]) * 1024, float(w["WRITE_SIZE"]) * 1024
    hit, miss = float(l2["TCC_HIT_sum"]), float(l2["TCC_MISS_sum"])
    us = float(f["us"])
    roof = max(flops / PEAK, alg / HBM)
    out.append({"name": name, "kernel": f["kernel"], "what": desc, "blocks": int(f["blocks"]), "launch_us_profiled": us,
                "bound": "mfma" if flops / PEAK >= alg / HBM else "hbm", "flops": flops, "algorithmic_bytes": alg,
                "traffic_bytes": int(fetch + write), "fetch_bytes_corrected": int(fetch), "write_bytes": int(write),
                "l2_hit_rate": round(hit / (hit + miss), 4), "roof_us": round(roof * 1e6, 1), "frac_of_mixed_roof": round(roof * 1e6 / us, 3),
                "achieved_tflops": round(flops / us / 1e6, 1), "achieved_gbs_algorithmic": round(alg / us / 1e3, 1), "commit": commit,
                "source": f"profiles/{tag}_pmc_family.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_family_bl.sh at commit {commit})"})
json.dump(out, open(os.path.join(O, f"{tag}_pmc_family.json"), "w"), indent=1)
for r in out:
    print(r["name"], r["launch_us_profiled"], "us  traffic", r["traffic_bytes"] / 1e6, "MB  algorithmic", r["algorithmic_bytes"] / 1e6, "MB  frac", r["frac_of_mixed_roof"])
