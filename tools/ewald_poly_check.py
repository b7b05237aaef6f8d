import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as Pn
f32=np.float32
alpha=2.6283; rl=1.1
r=np.linspace(0.08,1.1,200001)
exact=(erfc(alpha*r)/r+2*alpha/np.sqrt(np.pi)*np.exp(-(alpha*r)**2))/r**2     # force factor / qq
# current float path
def cur(r):
    r=r.astype(f32); r2=r*r; invR=f32(1)/np.sqrt(r2); a=f32(alpha)
    ar=(r2*invR)*a
    ex=np.exp2(r2*f32(-(alpha*alpha*1.4426950408889634))).astype(f32)
    tt=f32(1)/(ar*f32(0.3275911)+f32(1))
    poly=tt*f32(1.061405429)+f32(-1.453152027); poly=poly*tt+f32(1.421413741); poly=poly*tt+f32(-0.284496736); poly=poly*tt+f32(0.254829592)
    erfcv=poly*tt*ex
    f=(invR)*(erfcv+(ar*ex)*f32(1.1283791670955126))
    return (f*(invR*invR)).astype(np.float64)
# polynomial g(z): F = 1/r^3 - alpha^3 g(z)
zmax=(alpha*rl)**2
def g(z):
    z=np.asarray(z,dtype=np.float64); out=np.empty_like(z); s=z<1e-3
    zs=z[~s]; out[~s]=(erf(np.sqrt(zs))/np.sqrt(zs)-2/np.sqrt(np.pi)*np.exp(-zs))/zs
    zz=z[s]; out[s]=4/(3*np.sqrt(np.pi))*(1-3*zz/5+3*zz*zz/14-zz**3/27)
    return out
x=np.cos(np.pi*(np.arange(4000)+0.5)/4000); zz=(x+1)/2*zmax
for deg in (10,11,12):
    c=C.chebfit(x,g(zz),deg)
    mono=C.cheb2poly(c)     # monomial in t in [-1,1]
    def pol(r):
        r=r.astype(f32); r2=r*r; invR=f32(1)/np.sqrt(r2)
        t=r2*f32(2*alpha*alpha/zmax)-f32(1)
        acc=np.full_like(t,f32(mono[-1]))
        for k in range(len(mono)-2,-1,-1): acc=acc*t+f32(mono[k])
        invR3=invR*invR*invR
        return (invR3-f32(alpha**3)*acc).astype(np.float64)
    e=pol(r)
    print("deg",deg,"max abs err*r (force units per qq):",np.max(np.abs(e-exact)*r),"max rel",np.max(np.abs(e-exact)/exact), "coef max",np.abs(mono).max())
e=cur(r)
print("current: max abs err*r:",np.max(np.abs(e-exact)*r),"max rel",np.max(np.abs(e-exact)/exact))
# error near cutoff where exact is small
m=r>0.9
for name,fn in (("cur",cur),):
    print(name,"near cutoff max abs*r", np.max(np.abs(fn(r[m])-exact[m])*r[m]))
