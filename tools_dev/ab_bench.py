# developer helper: bench.py against another build of the library (WMX_TOOL_LIB=path), for same-box A/B runs
import os, runpy, sys
sys.path.insert(0, os.getcwd())
from wmix_amd import _lib
if os.environ.get('WMX_TOOL_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['WMX_TOOL_LIB'])
sys.argv = ['bench.py'] + sys.argv[1:]
runpy.run_path('bench.py', run_name='__main__')
