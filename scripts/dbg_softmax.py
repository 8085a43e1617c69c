import sys; sys.path.insert(0,'.')
import numpy as np, __graft_entry__ as ge
B=ge.load_package().binding; O=ge.load_oracle()
x=np.array([1,2,3,4],np.float32)
g=B.softmax(x); r=O.softmax(x)
print("gpu", [v.hex() for v in g.astype(float)]); print("orc", [v.hex() for v in r.astype(float)])
