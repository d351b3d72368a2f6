"""__graft_entry__.smoke() as a gpu_call.sh step: `py:tools/r06/smoke.py`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.smoke()
print("smoke OK")
