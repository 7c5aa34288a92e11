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
    old = "      const uint32_t* gmx_aw = reinterpret_cast<const uint32_t*>(A.ancestors_d);                 \\\n"
    assert old in src
    new = ("      if (true) { _Pragma(\"unroll\") for (int p = 0; p < PP; ++p) arow[p] = cidx[p]; } else {           \\\n" + old)
    src = src.replace(old, new, 1)
    old2 = "        arow[p] = gmx_i < n32 ? gmx_i : n32 - 1u;                                                \\\n      }                                                                                          \\\n"
    assert old2 in src
    src = src.replace(old2, old2 + "      }                                                                                          \\\n", 1)
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
