"""Runs tools/wf_one.py under several builds of the library (RT1W_LIB): python tools/wf_variants.py <arm> <W> <H> <spp> lib1 lib2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:5]
for lib in sys.argv[5:]:
    env = dict(os.environ, RT1W_LIB=os.path.join(ROOT, "raytracing-1w_amd", lib))
    print("==", lib, flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wf_one.py"), *args], env=env, check=False, timeout=200)
