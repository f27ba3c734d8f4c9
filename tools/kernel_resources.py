"""Per-kernel register / scratch / LDS usage of the gfx950 code objects: compiles each csrc/*.hip with
-Rpass-analysis=kernel-resource-usage (no GPU needed) and prints one line per kernel; exit code 1 if any kernel spills
(non-zero scratch).  Usage: python tools/kernel_resources.py [file stem ...] [--min-vgpr N]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "vibravox_amd", "csrc")


def analyse(stem, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CS}", "-DEBEN_BUILDING=1",
               "-c", os.path.join(CS, stem + ".hip") if os.path.exists(os.path.join(CS, stem + ".hip")) else os.path.join(CS, "exact_fp32", stem + ".hip"), "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage", *extra]
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split()[0].strip()
        def g(k):
            m = re.search(re.escape(k) + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn).replace("void eben::", "")
        out.append(dict(file=stem, kernel=dn, vgpr=g("VGPRs"), agpr=g("AGPRs"), sgpr=g("SGPRs"), scratch=g("ScratchSize [bytes/lane]"),
                        occ=g("Occupancy [waves/SIMD]"), lds=g("LDS Size [bytes/block]")))
    return out


if __name__ == "__main__":
    stems = [a for a in sys.argv[1:] if not a.startswith("-")] or sorted(f[:-4] for d in (CS, os.path.join(CS, "exact_fp32")) for f in os.listdir(d) if f.endswith(".hip"))
    bad = 0
    for s in stems:
        for k in analyse(s):
            flag = "  <-- SPILLS" if k["scratch"] > 0 else ""
            bad += k["scratch"] > 0
            print(f"{k['file']:10s} {k['kernel']:60s} vgpr {k['vgpr']:4d} agpr {k['agpr']:4d} sgpr {k['sgpr']:4d} scratch {k['scratch']:4d} occ {k['occ']} lds {k['lds']}{flag}")
    sys.exit(1 if bad else 0)
