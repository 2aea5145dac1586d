# Disassembles the small-tile GEMM instantiation and prints instruction-class counts per region (GPU box: fast compile)
set -e
cd $GRAFT_REPO_ROOT/image-captioning-with-external-knowledge_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -I ../../include -I . -S --cuda-device-only gemm.hip -o $GRAFT_REPO_ROOT/gpurun_out/gemm.s
python3 - <<'PY'
import re,os
s=open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/gemm.s").read()
# find the kernel gemm_kernel<2,2,1,1,false,false,true,32>
m=re.search(r"^(_ZN3ick[^\n:]*gemm_kernelILi2ELi2ELi1ELi1ELb0ELb0ELb1ELi32E[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm", s, re.S|re.M)
if m is None:
    print([l for l in s.splitlines() if "gemm_kernelILi2ELi2ELi1ELi1ELb0ELb0ELb1" in l][:5]); raise SystemExit(1)
body=m.group(2).splitlines()
import collections
c=collections.Counter()
for l in body:
    l=l.strip()
    if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"): continue
    op=l.split()[0]
    cls = "mfma" if "mfma" in op else ("ds" if op.startswith("ds_") else ("buffer/global" if op.startswith(("buffer_","global_","flat_")) else ("valu" if op.startswith("v_") else ("salu" if op.startswith("s_") else "other"))))
    c[cls]+=1
print("static instruction counts:", dict(c), "total", sum(c.values()))
ops=collections.Counter(l.strip().split()[0] for l in body if l.strip() and not l.strip().startswith((";",".")) and not l.strip().endswith(":"))
print(ops.most_common(25))
PY
