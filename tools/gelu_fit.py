import numpy as np
from scipy.special import erf
def Phi(x): return 0.5*(1+erf(x/np.sqrt(2)))
def phi(x): return np.exp(-x*x/2)/np.sqrt(2*np.pi)
def gelu(x): return x*Phi(x)
def dgelu(x): return Phi(x)+x*phi(x)
def fit_odd(f, c, deg, wfun, iters=200):
    n = 6000
    x = c*np.cos(np.pi*(np.arange(n)+0.5)/(2*n))
    s = x*x
    V = np.vander(s/(c*c), deg, increasing=True) * x[:,None]
    y = f(x); w0 = wfun(x)
    wt = np.ones(n)
    for it in range(iters):
        A = V*(wt*w0)[:,None]; b = y*wt*w0
        co, *_ = np.linalg.lstsq(A, b, rcond=None)
        e = np.abs(V@co - y)*w0
        wt = wt*(0.3+e/e.max()); wt/=wt.max()
    return co / (c*c)**np.arange(deg)
xs = np.linspace(-12,12,480001)
def ev32(co, x32, c):
    xc = np.clip(x32,-np.float32(c),np.float32(c)); s=(xc*xc).astype(np.float32)
    P = np.zeros_like(s)
    for k in co.astype(np.float32)[::-1]: P = (P*s+k).astype(np.float32)
    return (np.float32(0.5) + xc*P).astype(np.float32)
x32 = xs.astype(np.float32)
for c, deg in ((4.5,9),(4.5,10),(4.75,10),(5.0,10),(5.0,11)):
    co = fit_odd(lambda x: Phi(x)-0.5, c, deg, lambda x: 1+ x*x)
    ph = ev32(co, x32, c)
    y = (x32*ph).astype(np.float32)
    e = np.abs(y.astype(np.float64)-gelu(xs))
    m = (e/(2.0**-9*np.abs(gelu(xs))+1e-4)).max()
    print(f"PHI c={c} terms={deg}: max|Phi err|={np.abs(ph-Phi(xs)).max():.2e} abs gelu err={e.max():.2e} metric={m:.3f}")
    print("   ", ", ".join(f"{v:.9e}" for v in co))
for c, deg in ((4.5,6),(4.5,7),(4.5,8),(5.0,8)):
    co = fit_odd(lambda x: dgelu(x)-0.5, c, deg, lambda x: 1+0*x)
    d = ev32(co, x32, c)
    e = np.abs(d.astype(np.float64)-dgelu(xs))
    print(f"DG  c={c} terms={deg}: max err={e.max():.2e}")
    print("   ", ", ".join(f"{v:.9e}" for v in co))
