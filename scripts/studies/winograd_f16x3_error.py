"""CPU study for the "next" list of DESIGN.md section 8: would Winograd F(2x2,3x3) on the f16x3 operands stay inside the
engine's precision budget?  Emulates, for a 3x3 / stride 1 / pad 1 convolution with the decoder's shapes,
  (a) torch fp32 direct,  (b) direct f16x3 (x = hi + lo, w = hi + lo, hi*hi + hi*lo + lo*hi in fp32),
  (c) Winograd F(2x2,3x3) with the TRANSFORMED input tiles and the transformed weights split hi/lo, the 16 per-position
      channel GEMMs in f16x3 arithmetic, input / output transforms in fp32
against an fp64 direct convolution.  Prints max / rms error relative to the output rms.
   python scripts/studies/winograd_f16x3_error.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
torch.set_num_threads(16)


def split(x):
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return hi.float(), lo.float()


def mm3(a, b):
    """a @ b with both operands hi/lo split, three fp32 products (what three fp16 MFMAs with fp32 accumulate compute)."""
    ah, al = split(a)
    bh, bl = split(b)
    return al @ bh + ah @ bl + ah @ bh


def direct_f16x3(x, w):
    n, c, h, wd = x.shape
    cols = F.unfold(x, 3, padding=1)                                  # [n, c*9, h*w]
    wm = w.reshape(w.shape[0], -1)
    scale = 2.0 ** (14 - torch.floor(torch.log2(wm.abs().max())))     # the engine's power-of-two weight pre-scaling
    out = torch.stack([mm3(wm * scale, cols[i]) for i in range(n)]) / scale
    return out.view(n, w.shape[0], h, wd)


BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def winograd_f16x3(x, w):
    n, c, h, wd = x.shape
    k = w.shape[0]
    U = torch.einsum("ij,kcjl,ml->imkc", G.double(), w.double(), G.double()).float()          # [4,4,K,C] (offline, fp64 -> fp32)
    scale = 2.0 ** (14 - torch.floor(torch.log2(U.abs().max())))
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # [n, c, th, tw, 4, 4]
    V = torch.einsum("ij,ncabjl,ml->imncab", BT, tiles, BT)           # fp32 input transform  [4,4,n,c,th,tw]
    th, tw = V.shape[-2:]
    M = torch.empty(4, 4, n, k, th, tw)
    for i in range(4):
        for j in range(4):
            v = V[i, j].permute(1, 0, 2, 3).reshape(c, -1)            # [c, n*th*tw]
            M[i, j] = (mm3(U[i, j] * scale, v) / scale).view(k, n, th, tw).permute(1, 0, 2, 3)
    Y = torch.einsum("ij,jlnkab,ml->nkaibm", AT, M, AT)               # fp32 output transform [n,k,th,2,tw,2]
    return Y.reshape(n, k, 2 * th, 2 * tw)[:, :, :h, :wd]


def report(name, y, ref):
    e = (y.double() - ref).abs()
    r = ref.pow(2).mean().sqrt()
    print(f"  {name:34s} max {float(e.max() / r):.2e}   rms {float(e.pow(2).mean().sqrt() / r):.2e}   (of the output rms)")


for cin, cout, hw, relu in ((256, 256, 32, True), (512, 256, 24, True), (1024, 512, 16, False)):
    x = torch.randn(1, cin, hw, hw) * 2
    if relu:
        x = torch.relu(x)
    w = torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    print(f"3x3 conv {cin} -> {cout} on {hw}x{hw}{' (ReLU input)' if relu else ''}")
    report("torch fp32 direct", F.conv2d(x, w, padding=1), ref)
    report("f16x3 direct (the engine)", direct_f16x3(x, w), ref)
    report("f16x3 Winograd F(2x2,3x3)", winograd_f16x3(x, w), ref)
