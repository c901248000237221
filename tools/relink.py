"""Debug aid: rebuild ONE object with extra flags and relink (python tools/relink.py elementwise.hip -DX)."""
import subprocess, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yoloret_amd import build as B
os.chdir(B.HERE)
src = sys.argv[1]
subprocess.check_call(['/opt/rocm/bin/hipcc'] + B.FLAGS + sys.argv[2:] + ['-c', 'csrc/' + src, '-o', 'csrc/_obj/' + src.replace('.hip', '.o')], stderr=subprocess.DEVNULL)
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', B.LIB] + [os.path.join('csrc/_obj', f.replace('.hip', '.o')) for f in B.SOURCES])
print('relinked')
