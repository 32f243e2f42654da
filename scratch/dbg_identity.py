import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import golden_case_inputs, load_golden
from test_reproj_gpu import run_hip
c = golden_case_inputs("identity_pose")
r = run_hip(c, g=(1.0, 0.0), want_minmap=True)
print("losses", r["losses"])
for i in range(3):
    m = r["minmap"][i]
    print(i, "minmap max", m.max(), "nonzero frac", (m != 0).mean(), "d_inv max", np.abs(r["d_inv"][i]).max())
print("d_pose", r["d_pose"])
