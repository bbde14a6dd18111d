import sys, ctypes; sys.path.insert(0,'.')
import numpy as np
from dosma_amd import _lib as L
L._SO = 'scripts/libqmri_trace.so'
y=np.array([260.05743 , 174.9708  , 117.72315 ,  79.206024,  53.291077, 35.85509 ,  24.123878,  16.23093 ],dtype=np.float32).reshape(8,1)
x=np.arange(1,9)*10.0
o=L.monoexp_fit_host(x,y,p0=(1.0,-1/30),want_info=True)
print(o)
