import numpy as np
from math import sqrt
rng=np.random.default_rng(0)

def old_block(s):
    s00,s10,s20,s30,s11,s21,s31,s22,s32,s33 = s
    r0=1/sqrt(s00); l10=s10*r0; l20=s20*r0; l30=s30*r0
    p1=s11-l10*l10; r1=1/sqrt(p1)
    l21=(s21-l20*l10)*r1; l31=(s31-l30*l10)*r1
    p2=(s22-l20*l20)-l21*l21; r2=1/sqrt(p2)
    l32=((s32-l30*l20)-l31*l21)*r2
    p3=((s33-l30*l30)-l31*l31)-l32*l32; r3=1/sqrt(p3)
    y10=-r1*(l10*r0); y21=-r2*(l21*r1); y32=-r3*(l32*r2)
    y20=-r2*(l21*y10+l20*r0); y31=-r3*(l32*y21+l31*r1)
    y30=-r3*(l32*y20+(l31*y10+l30*r0))
    return np.array([[r0,0,0,0],[y10,r1,0,0],[y20,y21,r2,0],[y30,y31,y32,r3]])

def new_block(s):
    s00,s10,s20,s30,s11,s21,s31,s22,s32,s33 = s
    t11=s00*s11-s10*s10; t21=s00*s21-s20*s10; t31=s00*s31-s30*s10
    t22=s00*s22-s20*s20; t32=s00*s32-s30*s20; t33=s00*s33-s30*s30
    u22=t11*t22-t21*t21; u32=t11*t32-t31*t21; u33=t11*t33-t31*t31
    w33=u22*u33-u32*u32
    q0=1/sqrt(s00); q1=1/sqrt(t11); q2=1/sqrt(u22); q3=1/sqrt(w33)
    rho1=q0*q1; rho2=rho1*q2; rho3=rho2*q3
    a=t11*s00
    R20=t21*s10-t11*s20; R21=-(t21*s00)
    R3a=t31*s10-t11*s30; R3b=-(t31*s00)
    R30=u22*R3a-u32*R20; R31=u22*R3b-u32*R21; R32=-(u32*a); R33=u22*a
    return np.array([[q0,0,0,0],[-s10*rho1,s00*rho1,0,0],[R20*rho2,R21*rho2,a*rho2,0],[R30*rho3,R31*rho3,R32*rho3,R33*rho3]])

def pack(S): return [S[0,0],S[1,0],S[2,0],S[3,0],S[1,1],S[2,1],S[3,1],S[2,2],S[3,2],S[3,3]]

for cond in [1e1,1e4,1e8,1e12]:
    eo=en=0
    for it in range(2000):
        Q,_=np.linalg.qr(rng.normal(size=(4,4)))
        ev=np.exp(rng.uniform(0,np.log(cond),4)); S=(Q*ev)@Q.T; S=(S+S.T)/2
        Yo=old_block(pack(S)); Yn=new_block(pack(S))
        # residual Y S Y^T - I
        eo=max(eo,np.abs(Yo@S@Yo.T-np.eye(4)).max()); en=max(en,np.abs(Yn@S@Yn.T-np.eye(4)).max())
    print(cond,eo,en)
