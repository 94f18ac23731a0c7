"""THIS CONTAINER ONLY (needs oracle/_ref): random images with a side over 1024 px - any aspect ratio down to 1 px, 1 / 3 / 4 channels, noise /
gradients / blocks - through the reference's Image::Load (stb_image_resize 0.97, Mitchell) and this repository's: the reduced RGBA8 must be
bit-identical.   python3 tools/fuzz_resize_vs_reference.py [first_seed] [count]     (round 3: 60 images, 0 mismatches)"""
import sys, os, numpy as np, tempfile, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle.ref_binding import Ref
from pbrpathtracer_amd import pathtracer as P
from test_host_cpu import _write_png
ref=Ref(); tmp=tempfile.mkdtemp(); bad=0
first=int(sys.argv[1]) if len(sys.argv) > 1 else 0; count=int(sys.argv[2]) if len(sys.argv) > 2 else 20
for seed in range(first, first+count):
    rng=np.random.default_rng(seed)
    big=int(rng.integers(1025, 2600)); small=int(rng.integers(1, 1500)) if rng.uniform()<0.7 else int(rng.integers(1025,2000))
    w,h=(big,small) if rng.uniform()<0.5 else (small,big)
    ch=int(rng.choice([1,3,4]))
    kind=rng.uniform()
    if kind<0.4: a=rng.integers(0,256,(h,w,ch),dtype=np.uint8)
    elif kind<0.8:
        yy,xx=np.mgrid[0:h,0:w]; a=np.stack([((xx*7+yy*3)%256),((xx//5+yy)%256),((xx*yy)%256),((xx+yy*2)%256)],-1)[...,:ch].astype(np.uint8)
    else: a=np.repeat(np.repeat(rng.integers(0,256,((h+15)//16,(w+15)//16,ch),dtype=np.uint8),16,0),16,1)[:h,:w]
    if ch==1: a=a[...,0]
    p=os.path.join(tmp,'r.png'); _write_png(p,a)
    ww=C.c_int(); hh=C.c_int(); ok=ref.lib.ref_image_load(p.encode(), C.byref(ww), C.byref(hh))
    want=None
    if ok==1:
        want=np.zeros((hh.value,ww.value,4),np.uint8); ref.lib.ref_image_data(want.ctypes.data_as(C.POINTER(C.c_ubyte)))
    got=P.image_load(p)
    same=(want is None and got is None) or (want is not None and got is not None and want.shape==got.shape and np.array_equal(want,got))
    if not same:
        bad+=1; print("MISMATCH seed",seed,(w,h,ch),None if want is None else want.shape,None if got is None else got.shape, '' if (want is None or got is None or want.shape!=got.shape) else ('max diff %d, differing %d'%(np.abs(want.astype(int)-got.astype(int)).max(), int((want!=got).sum()))), flush=True)
print("images",count,"mismatches",bad)
