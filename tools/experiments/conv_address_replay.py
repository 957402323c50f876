"""Host-side replay of the address arithmetic of k_conv_gemm with unconditional clamped loads (see the patch beside
this file): every load of every thread, block and K-step (including the prefetches past the end) is checked against the
tensor sizes, for the test shapes of tests/test_gpu_conv.py."""
import numpy as np, itertools
BK=32
def check(MODE, MT, NT, WN, NI,Hs,Ws,cs,cb, grid, kps=0):
    BM = 32*MT*(4//WN); BN = 32*NT*WN
    RA = BM//64; RB = BN//64 if BN>=64 else 1
    Hb,Wb=2*Hs,2*Ws; pixels=NI*Hs*Ws
    nT=NI*Hb*Wb*cb; nS=NI*Hs*Ws*cs; nQ=cs*16*cb
    tid=np.arange(256)
    bad=[]
    for bx in range(grid[0]):
      for by in range(grid[1]):
        for bz in range(grid[2]):
            m0=by*BM; n0=bx*BN
            py=px=0
            if MODE==1: py=bz>>1; px=bz&1
            if MODE==0: nk=16*cb//BK; ks0=0
            elif MODE==1: nk=4*cs//BK; ks0=0
            else:
                nk_all=(pixels+BK-1)//BK; ks0=bz*kps; nk=min(ks0+kps,nk_all)-ks0
            if nk<=0: continue
            # rows bookkeeping
            pn=[];pa=[];pb=[];pv=[]
            for i in range(2*RA):
                m=m0+(tid>>3)+32*i
                v=m<pixels; mm=np.where(v,m,0)
                n=mm//(Hs*Ws); r=mm-n*(Hs*Ws); a=r//Ws; b=r-a*Ws
                pn.append(n);pa.append(a);pb.append(b);pv.append(v)
            for ksreq in range(0, nk+3):
                kst=min(ksreq,nk-1)
                # A
                if MODE==0:
                    k0=kst*BK; cell=k0//cb; c0=k0-cell*cb+(tid&7)*4; kh=cell>>2; kw=cell&3
                    for i in range(2*RA):
                        y=2*pa[i]-1+kh; x=2*pb[i]-1+kw
                        ok=pv[i]&(y>=0)&(y<Hb)&(x>=0)&(x<Wb)
                        yc=np.where(ok,y,0); xc=np.where(ok,x,0)
                        off=((pn[i]*Hb+yc)*Wb+xc)*cb+c0
                        if (off<0).any() or (off+4>nT).any(): bad.append(('A down',bx,by,bz,ksreq,i,off.min(),off.max(),nT))
                elif MODE==1:
                    k0=kst*BK; tap=k0//cs; s0=k0-tap*cs+(tid&7)*4; th=tap>>1; tw=tap&1
                    da=(0 if th else 1) if py else (-1 if th else 0); db=(0 if tw else 1) if px else (-1 if tw else 0)
                    for i in range(2*RA):
                        a=pa[i]+da; b=pb[i]+db
                        ok=pv[i]&(a>=0)&(a<Hs)&(b>=0)&(b<Ws)
                        ac=np.where(ok,a,0); bc=np.where(ok,b,0)
                        off=((pn[i]*Hs+ac)*Ws+bc)*cs+s0
                        if (off<0).any() or (off+4>nS).any(): bad.append(('A up',bx,by,bz,ksreq,i,off.min(),off.max(),nS))
                else:
                    for i in range(2*RA):
                        j=m0+(tid&15)*4+64*(i>>1); m=(ks0+kst)*BK+(tid>>4)+16*(i&1)
                        off=np.minimum(m,pixels-1)*cs+np.minimum(j,cs-4)
                        if (off<0).any() or (off+4>nS).any(): bad.append(('A wgrad',bx,by,bz,ksreq,i,off.min(),off.max(),nS))
                # B
                if MODE==0:
                    k=kst*BK+(tid&7)*4
                    for i in range(2*RB):
                        j=n0+(tid>>3)+32*i
                        off=np.minimum(j,cs-1)*16*cb+k
                        if (off<0).any() or (off+4>nQ).any(): bad.append(('B down',bx,by,bz,ksreq,i,off.min(),off.max(),nQ))
                elif MODE==1:
                    k0=kst*BK; tap=k0//cs; s0=k0-tap*cs; th=tap>>1; tw=tap&1
                    kh=(2 if th else 0) if py else (3 if th else 1); kw=(2 if tw else 0) if px else (3 if tw else 1)
                    for i in range(2*RB):
                        c=n0+(tid&15)*4+64*(i>>1); s_=s0+(tid>>4)+16*(i&1)
                        off=(s_*16+kh*4+kw)*cb+np.minimum(c,cb-4)
                        if (off<0).any() or (off+4>nQ).any(): bad.append(('B up',bx,by,bz,ksreq,i,off.min(),off.max(),nQ))
                else:
                    for q in range(RB):
                        col=n0+(tid&15)*4+64*q
                        nv=col<16*cb
                        colc=np.where(nv,col,0); cell=colc//cb; wc=colc-cell*cb; wkh=cell>>2; wkw=cell&3
                        for i in range(2):
                            m=(ks0+kst)*BK+(tid>>4)+16*i
                            inn=m<pixels; mm=np.where(inn,m,0)
                            n=mm//(Hs*Ws); r=mm-n*(Hs*Ws); a=r//Ws; b=r-a*Ws
                            y=2*a-1+wkh; x=2*b-1+wkw
                            ok=inn&nv&(y>=0)&(y<Hb)&(x>=0)&(x<Wb)
                            yc=np.where(ok,y,0); xc=np.where(ok,x,0)
                            off=((n*Hb+yc)*Wb+xc)*cb+wc
                            if (off<0).any() or (off+4>nT).any(): bad.append(('B wgrad',bx,by,bz,ksreq,i,q,off.min(),off.max(),nT))
            if len(bad)>5: return bad
    return bad
def splits(pixels,cs,cb):
    tiles=((cs+63)//64)*((16*cb+127)//128); nk=(pixels+31)//32; s=1
    while tiles*s<512 and s*2<=64 and nk//(s*2)>=8: s*=2
    return s
LAYERS=[(20,16,16,64,32),(20,8,8,128,64),(20,4,4,256,128),(3,5,7,32,32),(2,1,1,64,96),(70,2,3,36,64)]
for (NI,Hs,Ws,cs,cb) in LAYERS:
    pixels=NI*Hs*Ws
    if cb%32==0:
        b=check(0,2,1,2,NI,Hs,Ws,cs,cb,((cs+63)//64,(pixels+127)//128,1)); print('down',(NI,Hs,Ws,cs,cb),b[:2])
    if cs%32==0:
        if cb<=32: b=check(1,2,1,1,NI,Hs,Ws,cs,cb,(1,(pixels+255)//256,4))
        else: b=check(1,2,1,2,NI,Hs,Ws,cs,cb,((cb+63)//64,(pixels+127)//128,4))
        print('up',(NI,Hs,Ws,cs,cb),b[:2])
    sp=splits(pixels,cs,cb); nk=(pixels+31)//32
    b=check(2,1,2,2,NI,Hs,Ws,cs,cb,((16*cb+127)//128,(cs+63)//64,sp),(nk+sp-1)//sp); print('wgrad',(NI,Hs,Ws,cs,cb),sp,b[:2])
