import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo'); sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
import pathlib, test_cpp_shim as t
p, *_ = t._scene(pathlib.Path(sys.argv[1]))
print(p)
