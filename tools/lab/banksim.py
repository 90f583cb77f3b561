import numpy as np, sys

rng=np.random.default_rng(7)
def handmade(kind, edge):
    yy, xx = np.mgrid[0:edge, 0:edge]
    if kind=="smooth":
        out=[]
        for c in range(3):
            g = 60 + 50*c + 40*np.sin(xx/900.0 + c) + 30*np.cos(yy/700.0) + rng.integers(-2,3,(edge,edge))
            out.append(np.clip(g,0,255))
        return np.stack(out,-1).astype(np.uint8)
    if kind=="iid":
        return rng.integers(0,256,(edge,edge,3)).astype(np.uint8)
    if kind=="natural":
        # 1/f noise, three correlated channels + small sensor noise
        f=np.fft.fftfreq(edge)[:,None]**2+np.fft.fftfreq(edge)[None,:]**2; f[0,0]=1
        base=np.fft.ifft2(np.fft.fft2(rng.standard_normal((edge,edge)))/f**0.6).real
        base=(base-base.mean())/base.std()
        out=[]
        for c in range(3):
            extra=np.fft.ifft2(np.fft.fft2(rng.standard_normal((edge,edge)))/f**0.6).real
            extra=(extra-extra.mean())/extra.std()
            g=110+20*c+45*base+15*extra+rng.normal(0,1.5,(edge,edge))
            out.append(np.clip(np.rint(g),0,255))
        return np.stack(out,-1).astype(np.uint8)
    if kind=="steps":
        base = ((xx // 64) * 37 + (yy // 8) * 11) % 200
        return np.stack([(base + 10 * c) % 256 for c in range(3)], axis=-1).astype(np.uint8)
def cost(D):
    # D: [ninstr, 64] dword indices; cycles = max lanes per bank, computed for the whole wave and for two halves
    bank=D&31
    def maxload(b):
        cnt=np.zeros((b.shape[0],32),np.int32)
        for k in range(b.shape[1]):
            np.add.at(cnt,(np.arange(b.shape[0]),b[:,k]),1)
        return cnt.max(1)
    whole=maxload(bank).mean()
    halves=(maxload(bank[:,:32])+maxload(bank[:,32:])).mean()
    return whole, halves
edge=1024
for kind in ("iid","smooth","natural","steps"):
    t=handmade(kind,edge).reshape(-1,3)
    n=t[:,2].astype(np.int64)
    for stream,x in (("r",t[:,0].astype(np.int64)),("g",t[:,1].astype(np.int64))):
        # instruction j of a wave: lanes l -> pixel 256*w + 4*l + j
        npx=(len(n)//256)*256
        idx=np.arange(npx).reshape(-1,64,4)   # [wave-step][lane][j]
        res={}
        for name,fn in (
            ("xor n^2x", lambda n,x: ((x<<8)|(n^((x<<1)&0xFE)))>>1),
            ("lin n+5x rot", lambda n,x: (x<<7)|((n+5*x)&127)),
            ("lin n+10x", lambda n,x: ((x<<8)|((n+10*x)&255))>>1),
            ("lin n+6x", lambda n,x: ((x<<8)|((n+6*x)&255))>>1),
            ("lin n+14x", lambda n,x: ((x<<8)|((n+14*x)&255))>>1),
            ("lin n+22x", lambda n,x: ((x<<8)|((n+22*x)&255))>>1),
        ):
            D=fn(n[:npx],x[:npx])[idx]     # [ws][lane][j]
            D=D.transpose(0,2,1).reshape(-1,64)[:20000]
            res[name]=cost(D)
        print(kind,stream," ".join(f"{k}: {v[0]:.2f}/{v[1]:.2f}" for k,v in res.items()))
