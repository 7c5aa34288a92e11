import os, json
print(json.dumps({k: v for k, v in os.environ.items() if any(t in k for t in ("HIP", "HSA", "ROC", "AMD", "COMGR", "LLVM", "LD_PRELOAD", "OMP"))}, indent=0))
