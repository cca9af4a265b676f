"""Per-layer diagnostic of the HIP UNet against the oracle (prints, never asserts).  GPU box only."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffpir_amd
from oracle import unet_oracle as uo
from tests.gpu_common import make_model, rel_err


def main():
    e = diffpir_amd.Engine(0)
    prec = os.environ.get("DIFFPIR_PRECISION", "f32")
    e.set_precision(prec)
    print("precision", prec)
    for tag, hp, B, H, W in (("tiny", uo.tiny_hp(), 2, 32, 32), ("ffhq64", uo.ffhq_hp(), 1, 64, 64)):
        model, sd = make_model(e, hp)
        g = torch.Generator().manual_seed(3)
        x = torch.randn((B, 3, H, W), generator=g)
        t = torch.tensor([999, 37][:B])
        taps = {}
        ref = uo.unet_forward(sd, hp, x, t, taps=taps)
        out = e.unet_forward(e.to_device(x.numpy()), t.numpy()).numpy()
        print(f"== {tag}: output rel err {rel_err(out, ref.numpy()):.3e}  nan={np.isnan(out).any()}")
        for name, tv in taps.items():
            if name == "emb":
                continue
            try:
                got = e.read_tap(name).reshape(tv.shape)
                print(f"   {name:32s} {tuple(tv.shape)!s:22s} rel {rel_err(got, tv.numpy()):.3e}")
            except Exception as ex:
                print(f"   {name:32s} tap error {ex}")
    # timing of the FFHQ forward at 256x256
    hp = uo.ffhq_hp()
    model, sd = make_model(e, hp)
    for B in (1, 4, 16):
        x = e.to_device(np.random.default_rng(0).standard_normal((B, 3, 256, 256)).astype(np.float32))
        t = np.full(B, 500)
        out = e.unet_forward(x, t); e.sync()
        t0 = time.time()
        for _ in range(3):
            e.unet_forward(x, t, out=out)
        e.sync()
        dt = (time.time() - t0) / 3
        fl = e.unet_flops(256, 256) * B
        print(f"FFHQ fwd B={B}: {dt*1e3:.2f} ms  {fl/dt/1e12:.2f} TFLOP/s")
        e.prof_enable(True); e.prof_reset()
        e.unet_forward(x, t, out=out); e.sync()
        print("   prof:", {k: (round(v[0], 3), v[1]) for k, v in e.prof_read().items()})
        e.prof_enable(False)


if __name__ == "__main__":
    main()
