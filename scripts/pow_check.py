import numpy as np, torch, math
rng=np.random.RandomState(0)
x=np.abs(rng.randn(200000))*3+1e-6
ref=np.array([float(v)**0.6 for v in x])
npw=x**0.6
dev=torch.from_numpy(x).cuda().pow(0.6).cpu().numpy()
def ulps(a,b): return np.abs(a.view(np.int64)-b.view(np.int64))
print('numpy vs python: mismatch frac', (npw!=ref).mean(), 'max ulp', ulps(npw,ref).max())
print('device(torch/ocml) vs python: mismatch frac', (dev!=ref).mean(), 'max ulp', ulps(dev,ref).max())
# correctly rounded reference via mpmath-free: use decimal
from decimal import Decimal, getcontext
getcontext().prec=60
bad_py=0; bad_dev=0
for v,r,d in zip(x[:20000],ref[:20000],dev[:20000]):
    t=(Decimal(float(v)).ln()*Decimal('0.59999999999999997779553950749686919152736663818359375')).exp()
    cr=float(t)
    bad_py += (cr!=r); bad_dev += (cr!=d)
print('vs correctly rounded (20000): python misrounds', bad_py/20000, 'device', bad_dev/20000)
