"""Diagnostic: bench.py against another build of the library (ZKHIP_LIB=path), for same-box A/Bs of compile-time choices."""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zk_cryptography_amd import _native as N
if os.environ.get("ZKHIP_LIB"):
    N.LIB_PATH = os.path.abspath(os.environ["ZKHIP_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
