"""Builds a DIAGNOSTIC copy of libgenmi_hip.so whose specialised site programs skip part of the resample-first
prologue (timing bounds only: WRONG results by construction; never loaded by the product — a bench process names it
through GENMI_LIB).  Variants:
  nopoll   the workgroup does not wait for its own slots' ancestor words: it gathers through the identity instead
           (what removing the write-through + poll hop could save AT MOST, with the ancestor stores still issued)
Usage: python tools/experiments/build_diag_lib.py nopoll  -> tools/experiments/_alt/libgenmi_nopoll.so"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

variant = sys.argv[1]
work = f"/tmp/gmx_diag_{variant}"
shutil.rmtree(work, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "genjax_amd", "csrc"), os.path.join(work, "csrc"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
jit = os.path.join(work, "csrc", "gmx_jit.h")
src = open(jit).read()
if variant == "nopoll":
    mark = "#if defined(GMX_JIT_SH)\n// a SHARDED sweep's step"
    assert mark in src
    src = src.replace(mark, "#undef GMX_JIT_POLL_ANC\n#define GMX_JIT_POLL_ANC(TAG, STATUS, LIMIT) "
                            "_Pragma(\"unroll\") for (int p = 0; p < PP; ++p) arow[p] = cidx[p];\n" + mark, 1)
else:
    raise SystemExit("unknown variant")
open(jit, "w").write(src)
# the embedded headers (hiprtc compiles the specialised kernels from them)
names = [n for n, _ in ge.EMBED_HEADERS]
parts = ["// generated\n", f"#define GMX_EMBED_COUNT {len(names)}\n",
         "static const char* const gmx_embed_name[GMX_EMBED_COUNT] = {" + ", ".join(f'"{n}"' for n in names) + "};\n",
         "static const char* const gmx_embed_src[GMX_EMBED_COUNT] = {\n"]
for n in names:
    path = os.path.join(work, "include", n) if n == "genmi.h" else os.path.join(work, "csrc", n)
    parts.append('R"GMXEMB(' + open(path).read() + ')GMXEMB",\n')
parts.append("};\n")
open(os.path.join(work, "csrc", "gmx_embed.inc"), "w").write("".join(parts))
out_dir = os.path.join(ROOT, "tools", "experiments", "_alt")
os.makedirs(out_dir, exist_ok=True)
out = os.path.join(out_dir, f"libgenmi_{variant}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + ge.HIPCC_FLAGS + ["-I", os.path.join(work, "include"),
                                                                   os.path.join(work, "csrc", "gmx_kernels.hip"), "-o", out, "-lhiprtc"])
print(out)
