"""Register / scratch / occupancy table of one translation unit's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/diag/resusage.py ral_attnm.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ecg_denoise_amd", "csrc")
extra = os.environ.get("EXTRA", "").split()
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, cwd=d, capture_output=True, text=True)
cur = None; rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for n, r in rows.items():
    dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if flt and flt not in dn: continue
    print(f"{dn:50s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} "
          f"spillV {r.get('VGPRs Spill','?'):>3} spillS {r.get('SGPRs Spill','?'):>3} occ {r.get('Occupancy [waves/SIMD]','?')}")
