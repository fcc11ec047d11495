import numpy as np, sys
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
d = np.abs(a - b)
print(sys.argv[1], sys.argv[2], "max diff per channel", d.max(0), "n>1e-5", (d > 1e-5).sum(0))
w = np.argsort(-d.max(1))[:5]
for i in w: print(i, a[i], b[i])
