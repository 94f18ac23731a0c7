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
        # header bytes: sizes, depths, counts - the first 64 bytes, or where the format keeps its DIMENSIONS when that is further in
        # (Softimage PIC: width / height at offset 88; PSD: 14-21; BMP: 18-25; TGA: 12-15; GIF: 6-9; PNG IHDR: 16-23)
        dims={'pic':(88,92),'psd':(14,22),'bmp':(18,26),'tga':(12,16),'gif':(6,10),'png':(16,24)}
        lo,hi=next((v for k_,v in dims.items() if k_ in name.lower()),(0,64))
        if seed%8<4: lo,hi=0,64
        i=int(rng.integers(lo,min(hi,len(d)))); d[i]=int(rng.choice([0,255,127,128,1]))
    open(os.path.join(out,f"{seed}_{name}.bin"),"wb").write(bytes(d)); k+=1
print(k)
