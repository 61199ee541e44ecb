import sys; sys.path.insert(0,"/root/repo")
import importlib
rt=importlib.import_module("raytracing-1w_amd")
sc=rt.Scene.reference(5,build_seed=1)
print("key", sc.kernel_key())
import os; print(sorted(os.listdir(os.path.join(os.path.dirname(rt.LIB_PATH),"kernels"))))
ctx=rt.Context(sc,0)
print("specialised at create:", ctx.specialised())
try:
    print(ctx.specialise(cached_only=True))
except Exception as e: print("ERR", e)
try:
    print(ctx.specialise())
except Exception as e: print("ERR", e)
