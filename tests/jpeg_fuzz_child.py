"""child process of tests/test_jpeg_decode.py::test_host_half_survives_corrupt_files: mutated JPEG files (byte flips,
truncations, insertions; sequential, progressive, restart-marker and optimised-table seeds) through mi_jpeg_parse +
mi_jpeg_huffman - every file is either decoded or refused with an error code, the process must not crash."""
import sys, io, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from yolov7_d2_amd import _lib as L
lib=L.lib()
rng=np.random.RandomState(0)
def smooth(h,w):
    yy,xx=np.mgrid[0:h,0:w]
    base=np.stack([127+100*np.sin(xx/9.0+yy/17.0),127+100*np.cos(xx/13.0),127+100*np.sin(yy/7.0)],-1)
    return np.clip(base+rng.randint(-20,21,(h,w,3)),0,255).astype(np.uint8)
seeds=[]
for kw in (dict(quality=80,subsampling=2),dict(quality=80,subsampling=0,progressive=True),dict(quality=60,subsampling=1,restart_marker_blocks=2),dict(quality=90,subsampling=2,progressive=True,optimize=True)):
    buf=io.BytesIO(); Image.fromarray(smooth(48,72)).save(buf,format="JPEG",**kw); seeds.append(buf.getvalue())
n_ok=n_rej=0
for it in range(int(sys.argv[1])):
    d=bytearray(seeds[it%len(seeds)])
    for _ in range(rng.randint(1,6)):
        mode=rng.randint(3)
        if mode==0: d[rng.randint(len(d))]=rng.randint(256)
        elif mode==1 and len(d)>50: del d[rng.randint(20,len(d)):]   # truncate
        else:
            p=rng.randint(len(d)); d[p:p]=bytes(rng.randint(0,256,rng.randint(1,5)).astype(np.uint8))
    data=bytes(d)
    if len(data)<8: continue
    buf=(C.c_uint8*len(data)).from_buffer_copy(data)
    info=L.mi_jpeg_info()
    if lib.mi_jpeg_parse(buf,len(data),C.byref(info))!=0: n_rej+=1; continue
    if info.coef_count>50_000_000: n_rej+=1; continue
    coef=np.empty(info.coef_count,np.int16)
    rc=lib.mi_jpeg_huffman(buf,len(data),C.byref(info),coef.ctypes.data_as(C.c_void_p))
    n_ok+= rc==0; n_rej+= rc!=0
print("fuzz done ok",n_ok,"rejected",n_rej)
