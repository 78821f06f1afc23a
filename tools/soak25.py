"""2.5D half of tools/soak.py only (bisecting stream-order hazards): prints max |param| after 81 steps"""
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "fpl-plus_amd"))
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import importlib.util
src = open(os.path.join(os.getcwd(), "tools", "soak.py")).read().split('print("3D')[0]
exec(src)
print(os.environ.get("FPLX_JOIN_KIND", "-"), run([2, 2, 3, 3, 3], (4, 1, 28, 128, 128), 81)[1])
