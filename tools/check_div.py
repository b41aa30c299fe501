import numpy as np, torch
u = np.arange(256, dtype=np.uint8)
ref = u.astype(np.float32) / np.float32(255.0)
d = torch.from_numpy(u).cuda()
a = (d.to(torch.float32) / 255.0).cpu().numpy()
b = torch.div(d, 255.0).cpu().numpy()
c = (d.to(torch.float64) / 255.0).to(torch.float32).cpu().numpy()
print("f32 div mismatches", int((a != ref).sum()), "torch.div", int((b != ref).sum()), "f64 path", int((c != ref).sum()))
