"""HBM traffic record of the roofline kernel from the three PMC passes of tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE / TCC hit+miss,
each in its own rocprofv3 run): writes profiles/<tag>_pmc_melgan_l4_fwd_<math>.json, which bench.py reads for `roofline.traffic`.
FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950.  Usage: pmc_summary.py <tag> <math> <batch> <commit>"""
import json, os, subprocess, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag, math, batch, commit = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
match = "tap3_kernelILi4ELi5ELb0E" if math == "bf16" else "tap2_kernelILi4E"   # mangled: FM = 4 (bf16: the 5-unit tile prefetch = the forward launch of this layer)


def agg(db):
    out = subprocess.run([sys.executable, os.path.join(R, "tools", "rocpd_pmc.py"), db, "--match", match, "--agg"], capture_output=True, text=True).stdout.strip().splitlines()
    hdr = out[0].split(",")
    rows = [dict(zip(hdr, l.split(","))) for l in out[1:]]
    rows.sort(key=lambda r: -float(r["us"]))   # the forward launch is the longest of the family in this filter
    return rows[0]


f, w, l2 = (agg(os.path.join(R, "gpurun_out", f"pmc_{k}", "p_results.db")) for k in ("fetch", "write", "l2"))
fetch_kb, write_kb = float(f["FETCH_SIZE"]), float(w["WRITE_SIZE"])
hit, miss = float(l2["TCC_HIT_sum"]), float(l2["TCC_MISS_sum"])
cout, cin_g, k, l_in, l_out = 1024, 256, 41, 500, 125
alg = batch * 1024 * l_in * 4 + batch * cout * l_out * 4 + (cout * cin_g * k * (2 if math == "bf16" else 4))
rec = {
    "kernel": f"eben::{f['kernel']} MelGAN L4 forward (1024->1024, k41, s4, g4), {batch} items per launch (enhanced + reference), 500->125",
    "launch_batch": batch, "commit": commit,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum in separate passes (tools/pmc_traffic.sh), averages over the launches",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "l2_hit_rate": round(hit / (hit + miss), 4),
    "fetch_bytes_raw": int(fetch_kb * 1024), "fetch_bytes_corrected": int(2 * fetch_kb * 1024), "write_bytes": int(write_kb * 1024),
    "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + write_kb * 1024), "algorithmic_bytes_per_launch": alg,
    "launch_us_profiled": float(f["us"]),
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B for 16-B/lane loads: the LDS-DMA weight stream); the 4-B/lane "
            "input-tile loads are uncalibrated, so the true read traffic lies between the raw and the corrected figure.",
}
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
path = os.path.join(R, "gpurun_out", f"{tag}_pmc_melgan_l4_fwd_{math}.json")
json.dump(rec, open(path, "w"), indent=1)
print(path, rec["traffic_bytes_per_launch"], rec["algorithmic_bytes_per_launch"])
