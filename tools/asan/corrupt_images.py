import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
z1=np.load(ROOT+'/tests/golden/tier_k_images.npz'); z2=np.load(ROOT+'/tests/golden/tier_k_images_psd_pic.npz'); z3=np.load(ROOT+'/tests/golden/tier_k_images_jpeg_sampling.npz')
files=[]
for z in (z1,z2,z3):
    for n in z['names']: files.append((str(n), z['file_'+str(n)].tobytes()))
out=sys.argv[1]; os.makedirs(out,exist_ok=True); first=int(sys.argv[2]); count=int(sys.argv[3])
k=0
for seed in range(first, first+count):
    rng=np.random.default_rng(seed)
    name,data=files[seed%len(files)]
    d=bytearray(data); mode=seed%4
    if mode==0:
        for _ in range(int(rng.integers(1,6))): d[int(rng.integers(0,len(d)))]=int(rng.integers(0,256))
    elif mode==1: d=d[:int(rng.integers(1,len(d)))]
    elif mode==2:
        i=int(rng.integers(0,len(d))); d[i:i+int(rng.integers(1,16))]=bytes(rng.integers(0,256,int(rng.integers(1,16)),dtype=np.uint8))
    else:
        i=int(rng.integers(0,min(64,len(d)))); d[i]=int(rng.choice([0,255,127,128,1]))     # header bytes: sizes, depths, counts
    open(os.path.join(out,f"{seed}_{name}.bin"),"wb").write(bytes(d)); k+=1
print(k)
