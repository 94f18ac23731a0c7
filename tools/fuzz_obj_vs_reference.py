"""THIS CONTAINER ONLY (needs oracle/_ref built from /root/reference): random OBJ files - polygons of 3-12 corners, planar / grid / ring /
free vertices, relative indices, v / v/vt / v//vn / v/vt/vn corners, `g` `o` `s` `usemtl` `l` `p` statements in any order, degenerate faces -
through the reference's LoadObject (tinyobj 2.0.0) and through this repository's: the staged triangles must be bit-identical, in order.
   python3 tools/fuzz_obj_vs_reference.py [first_seed] [count]     (round 3: 4 000 files, 0 mismatches after two fixes)"""
import sys, os, numpy as np, tempfile, ctypes as C
sys.path.insert(0,'/root/repo')
from oracle.ref_binding import Ref, _fp
from pbrpathtracer_amd.pathtracer import PathTracer
ref=Ref(); tmp=tempfile.mkdtemp()
first=int(sys.argv[1]) if len(sys.argv) > 1 else 0; count=int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad=0
M=np.eye(4,dtype=np.float32).reshape(-1)
for seed in range(first, first+count):
    rng=np.random.default_rng(seed)
    nv=int(rng.integers(3,40))
    mode=int(rng.integers(0,5))
    if mode==0: P=rng.normal(0,1,(nv,3))
    elif mode==1: P=np.round(rng.normal(0,2,(nv,3)))            # grid: collinear / coincident points
    elif mode==2: P=np.c_[rng.normal(0,1,(nv,2)), np.zeros(nv)]  # planar z=0
    elif mode==3: P=np.c_[np.zeros(nv), rng.normal(0,1,(nv,2))]  # planar x=0
    else:
        ang=np.sort(rng.uniform(0,2*np.pi,nv)); r=rng.uniform(0.3,1.5,nv); P=np.c_[r*np.cos(ang), rng.normal(0,0.01,nv), r*np.sin(ang)]   # star-ish ring in xz
    lines=[("v %.6g %.6g %.6g"%tuple(p)) for p in P]
    has_n=rng.uniform()<0.5; has_t=rng.uniform()<0.5
    if has_n: lines+= ["vn %.4f %.4f %.4f"%tuple(rng.normal(0,1,3)) for _ in range(3)]
    if has_t: lines+= ["vt %.4f %.4f"%tuple(rng.uniform(0,1,2)) for _ in range(4)]
    nf=int(rng.integers(1,12))
    for f in range(nf):
        r=rng.uniform()
        if r<0.3: lines.append(rng.choice(["g grp%d"%f,"o obj%d"%f,"g","s %d"%int(rng.integers(0,4)),"s off","usemtl m%d"%f,"g a  b","l 1 2","p 1","f 1 2","l 1 2 3","o","p 1 2"]))
        if rng.uniform()<0.25: continue
        k=int(rng.choice([3,3,4,4,5,6,7,8,12]))
        k=min(k,nv)
        if mode==4 and rng.uniform()<0.7:
            start=int(rng.integers(0,nv)); idx=[(start+i)%nv for i in range(k)]
        else: idx=list(rng.choice(nv,k,replace=rng.uniform()<0.1))
        rel=rng.uniform()<0.2
        def corner(i):
            v = (i-nv) if rel else (i+1)
            s=str(v)
            if has_t and has_n: s+="/%d/%d"%(int(rng.integers(1,5)),int(rng.integers(1,4)))
            elif has_t: s+="/%d"%int(rng.integers(1,5))
            elif has_n: s+="//%d"%int(rng.integers(1,4))
            return s
        lines.append("f "+" ".join(corner(i) for i in idx))
    text="\n".join(lines)+"\n"
    p=os.path.join(tmp,"f.obj"); open(p,"w").write(text)
    ref.lib.ref_clear(); ref.lib.ref_load_obj(p.encode(), _fp(M)); t=ref.triangles()
    nel=ref.lib.ref_num_elements(0) if ref.lib.ref_num_objects() else -1
    pt=PathTracer(); pt.LoadObject(p, np.eye(4,dtype=np.float32))
    ok = pt.GetLoadedObjects()==([nel] if nel>=0 else []) and pt.GetTriangleCount()==len(t)
    if ok and len(t):
        s=pt.StagedScene()
        ok = np.array_equal(s['verts'],t[:,0:9]) and np.array_equal(s['normals'],t[:,9:18],equal_nan=True) and np.array_equal(s['uvs'],t[:,18:24]) and np.array_equal(s['tbn'].view(np.uint32),t[:,24:33].copy().view(np.uint32)) and np.array_equal(s['smoothing'],(t[:,33]!=0).astype(np.uint8)) and np.array_equal(s['material'],t[:,35].astype(np.int32))
    pt.close()
    if not ok:
        bad+=1; keep=os.path.join('/tmp',f'objfuzz_bad_{seed}.obj'); open(keep,'w').write(text); print("MISMATCH seed",seed,"mode",mode,"->",keep, "tris ref",len(t), flush=True)
print("files",count,"mismatches",bad)
