"""Run the reference's OWN restoration loop -- not a restatement of it (build container only).

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box: only the fixture generators (oracle/gen_golden*.py) and
tests that skip when the tree is absent call this.

The loop of the reference (`test_rho`, /root/reference/main_ddpir.py:249-536) is a closure nested in `main()`, so it cannot be
imported.  Round 1-4 re-typed its ~200 lines of glue by hand (oracle/live_reference.py) and round 4's review found what that
costs: the hand copy had lost the `sigma_ks` branch of main_ddpir.py:279-283 and the DPS_yt fixture was wrong by O(1).  This module
holds NO line of the loop.  It

  1. imports /root/reference/main_ddpir.py as a module (its top level is imports + class/function definitions);
  2. builds `config` with the reference's own `parse_args_and_config()` from a YAML file = the reference's configs/<task>.yaml with
     the case's overrides (so /255, `sigma`, `kernel_std`, the seeding ... are the reference's statements);
  3. takes, with `ast`, three pieces of `main()`'s body out of the source file and executes them unmodified, compiled under the
     reference's file name and line numbers:
       - the schedule statements (`betas = ...` up to the `t_start` branch, main_ddpir.py:184-200),
       - the `requires_grad` freeze (main_ddpir.py:236-239),
       - `def test_rho(config)` itself (main_ddpir.py:249-536),
     and, on request, the lambda/zeta sweep that calls it (main_ddpir.py:549-581);
  4. calls `test_rho(config)` with the free variables of the closure supplied as globals: `dataloader` (a list of batches in
     CustomDataset.__getitem__'s format), `model`, `diffusion`, `device`, `logger`, `test_results_ave`.

What is stubbed, and only this: `util.imsave_batch` (file output -> no-op), `util.tensor2uint_batch` (called on `x_0` at
main_ddpir.py:482: the argument is recorded, then the real function runs), `torch.randn_like` (routed to the case's seeded noise
function so that the engine can be fed the same draws in the same order), and optionally `utils_model.model_fn` /
`utils_model.grad_and_value` (real functions, results recorded for per-step traces).

`run_main()` goes one step further and runs the reference's `main()` itself end to end -- CustomDataset (image files, the shipped
kernels, degradation, noise), model construction, torch.load of a (synthetic) checkpoint, the sweep -- inside a scratch `cwd`.
"""
from __future__ import annotations

import ast
import contextlib
import importlib.util
import logging
import os
import subprocess
import sys
import tempfile
from collections import OrderedDict

import numpy as np
import torch
import yaml

from . import ref_import

MAIN_PY = os.path.join(ref_import.REF_ROOT, "main_ddpir.py")
CONDA_PY = "/opt/conda/bin/python3.9"          # has h5py (Levin09.mat is a v7.3 / HDF5 file)


# ------------------------------------------------------------------------------------------ stubs the reference's imports need
class _Cv2:
    """The three cv2 calls utils_image.imread_uint / imsave make, on PIL (cv2 is not installed here)."""
    IMREAD_UNCHANGED, COLOR_GRAY2RGB, COLOR_BGR2RGB = -1, 8, 4

    @staticmethod
    def imread(path, flag=1):
        from PIL import Image
        im = Image.open(path)
        if flag == 0:
            return np.array(im.convert("L"))
        a = np.array(im)
        return a if a.ndim == 2 else a[:, :, 2::-1].copy()          # cv2 hands back BGR

    @staticmethod
    def cvtColor(img, code):
        if code == _Cv2.COLOR_GRAY2RGB:
            return np.stack([img] * 3, axis=2)
        return img[:, :, ::-1].copy()

    @staticmethod
    def imwrite(path, img):
        return True


def _loadmat(path, *a, **k):
    """hdf5storage.loadmat for the reference's three kernel files: scipy reads the v5 ones, h5py (other interpreter) Levin09."""
    import scipy.io
    try:
        return scipy.io.loadmat(path)
    except NotImplementedError:
        out = tempfile.mktemp(suffix=".npz")
        code = ("import h5py, numpy as np, sys\n"
                "f = h5py.File(sys.argv[1], 'r'); refs = f['kernels']\n"
                "ks = [np.array(f[refs[i, 0]]).T for i in range(refs.shape[0])]\n"       # MATLAB is column-major: h5py sees the transpose
                "np.savez(sys.argv[2], **{f'k{i}': k for i, k in enumerate(ks)})\n")
        subprocess.run([CONDA_PY, "-c", code, path, out], check=True)
        z = np.load(out)
        cell = np.empty((1, len(z.files)), dtype=object)
        for i in range(len(z.files)):
            cell[0, i] = z[f"k{i}"]
        os.unlink(out)
        return {"kernels": cell}


_md = {}


def main_module():
    """/root/reference/main_ddpir.py imported as a module (nothing runs: its tail is guarded by __name__ == '__main__')."""
    if "m" in _md:
        return _md["m"]
    ref_import.load()                                   # sys.path + the empty stub modules
    cv2 = sys.modules["cv2"]
    if not hasattr(cv2, "imread"):
        for n in ("imread", "cvtColor", "imwrite", "IMREAD_UNCHANGED", "COLOR_GRAY2RGB", "COLOR_BGR2RGB"):
            setattr(cv2, n, getattr(_Cv2, n))
    h5 = sys.modules["hdf5storage"]
    if not hasattr(h5, "loadmat"):
        h5.loadmat = _loadmat
    spec = importlib.util.spec_from_file_location("reference_main_ddpir", MAIN_PY)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    _md["m"] = m
    return m


# ------------------------------------------------------------------------------------------ ast: pieces of main()'s body
def _names_assigned(node):
    out = set()
    if isinstance(node, ast.Assign):
        for t in node.targets:
            for n in ast.walk(t):
                if isinstance(n, ast.Name):
                    out.add(n.id)
    return out


def _mentions(node, text):
    return text in ast.unparse(node)


_pieces = {}


def main_pieces():
    """{'schedule': [stmts], 'freeze': [stmt], 'test_rho': [FunctionDef], 'sweep': [stmts]} straight from the source file."""
    if _pieces:
        return _pieces
    with open(MAIN_PY) as f:
        tree = ast.parse(f.read(), filename=MAIN_PY)
    main_fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    body = main_fn.body
    i_betas = next(i for i, n in enumerate(body) if "betas" in _names_assigned(n))
    i_logger = next(i for i, n in enumerate(body) if "logger_name" in _names_assigned(n))
    i_rho = next(i for i, n in enumerate(body) if isinstance(n, ast.FunctionDef) and n.name == "test_rho")
    freeze = [n for n in body if isinstance(n, ast.If) and _mentions(n.test, "DPS_y0") and _mentions(n, "requires_grad")]
    sweep = [n for n in body[i_rho + 1:] if isinstance(n, ast.If) and _mentions(n.test, "config.task") and _mentions(n, "test_rho(config)")]
    assert len(freeze) == 1 and len(sweep) == 1, (len(freeze), len(sweep))
    _pieces.update(schedule=body[i_betas:i_logger], freeze=freeze, test_rho=[body[i_rho]], sweep=sweep)
    _pieces["lines"] = {k: (v[0].lineno, v[-1].end_lineno) for k, v in _pieces.items()}
    _pieces["test_rho_deblur"] = [_DropQ1().apply(body[i_rho])]
    return _pieces


class _DropQ1(ast.NodeTransformer):
    """THE ONE EDIT, task 'deblur' only.  main_ddpir.py:302 `k_4d = torch.einsum('ab,cd->abcd', torch.eye(3), k_4d)` raises for the
    batched 3-D `k` the DataLoader hands over ("number of subscripts (2) does not match ... (3)"), and a 2-D `k`
    fails eleven lines later at np.expand_dims(k, 3) -- test_rho cannot run a deblur batch as shipped.  `k_4d` only feeds
    `degrade_op`, which only the first-order / DPS branches call (they are exercised on task 'sr'); the DiffPIR deblur path never
    reads it.  This transformer deletes that single statement and asserts it deleted exactly one; nothing else is touched."""

    def apply(self, fn):
        import copy
        self.n = 0
        out = ast.fix_missing_locations(self.visit(copy.deepcopy(fn)))
        assert self.n == 1, self.n
        return out

    def visit_Assign(self, node):
        if "k_4d" in _names_assigned(node) and _mentions(node.value, "torch.einsum('ab,cd->abcd'"):
            self.n += 1
            return None
        return node


def _exec(nodes, ns):
    exec(compile(ast.Module(body=list(nodes), type_ignores=[]), MAIN_PY, "exec"), ns)


# ------------------------------------------------------------------------------------------ helpers around the call
class _Proxy:
    """A module with a few attributes overridden (everything else is the reference's)."""

    def __init__(self, real, **over):
        self.__dict__["_real"], self.__dict__["_over"] = real, over

    def __getattr__(self, n):
        o = self.__dict__["_over"]
        return o[n] if n in o else getattr(self.__dict__["_real"], n)


@contextlib.contextmanager
def patched_randn_like(noise_fn):
    """Route every torch.randn_like (p_sample's and the driver's) through noise_fn, in call order."""
    if noise_fn is None:
        yield
        return
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: noise_fn(t)
    try:
        yield
    finally:
        torch.randn_like = orig


def build_unet(hp, sd, frozen=True):
    """The reference's UNetModel + diffusion for oracle.unet_oracle.UNetHP `hp` (script_util.create_model / create_gaussian_diffusion).
    frozen=False leaves requires_grad on, as main_ddpir.py:236-239 does for generate_mode DPS_y0."""
    ns = ref_import.load()
    model = ns.script_util.create_model(
        image_size=hp.image_size, num_channels=hp.model_channels, num_res_blocks=hp.num_res_blocks,
        channel_mult=",".join(str(int(c)) for c in hp.channel_mult) if hp.channel_mult else "",
        learn_sigma=hp.learn_sigma, class_cond=hp.class_cond, use_checkpoint=False,
        attention_resolutions=hp.attention_resolutions, num_heads=4,
        num_head_channels=hp.num_head_channels, num_heads_upsample=-1, use_scale_shift_norm=True,
        dropout=0.1, resblock_updown=True, use_fp16=False, use_new_attention_order=False)
    if hp.class_cond and hp.num_classes != 1000:
        model.label_emb = torch.nn.Embedding(hp.num_classes, 4 * hp.model_channels)
        model.num_classes = hp.num_classes
    model.load_state_dict(sd)
    model.eval()
    if frozen:
        for _, v in model.named_parameters():
            v.requires_grad = False
    diffusion = ns.script_util.create_gaussian_diffusion(steps=1000, learn_sigma=hp.learn_sigma)
    return model, diffusion


class _Labelled(torch.nn.Module):
    """The reference's driver never passes class labels (main_ddpir.py runs unconditional checkpoints); BASELINE config 5 is the
    class-conditional 512 topology, so the labels are bound to the network here and test_rho stays untouched."""

    def __init__(self, model, y):
        super().__init__()
        self.model, self.y = model, y

    def forward(self, x, t, **kw):
        return self.model(x, t, y=self.y, **kw)


def yaml_for(task, **over):
    """The reference's configs/<task>.yaml as a dict, with overrides.  LPIPS (needs the lpips package + VGG weights) and file
    output are switched off -- both are YAML keys of the reference, not edits."""
    name = {"deblur": "deblur", "sr": "sisr", "inpaint": "inpaint"}[task]
    with open(os.path.join(ref_import.REF_ROOT, "configs", f"{name}.yaml")) as f:
        d = yaml.safe_load(f)
    d.update(calc_LPIPS=False, save_L=False, save_E=False)
    d.update(over)
    return d


def reference_config(ydict, cwd):
    """config = the reference's parse_args_and_config() on a YAML file holding `ydict` (cwd -> scratch directory)."""
    md = main_module()
    ydict = dict(ydict, cwd=cwd)
    path = os.path.join(cwd, "opt.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(ydict, f)
    argv = sys.argv
    sys.argv = ["main_ddpir.py", "--opt", path]
    try:
        return md.parse_args_and_config()
    finally:
        sys.argv = argv


def run_test_rho(ydict, batches, model, diffusion, noise_fn=None, sweep=False, trace=None, direct=None, ns_over=None):
    """Execute the reference's test_rho on `batches`.

    ydict   : YAML dict (yaml_for(...)); `direct` = {attr: value} set on config AFTER parse_args_and_config -- the values the sweep
              would have put there (lambda_, zeta) when sweep=False.
    batches : list of (img_H u8 [B,H,W,C], img_L float [B,h,w,C] in [0,1], names, k [B,kh,kw], mask [B,H,W,C]) torch tensors,
              what DataLoader(CustomDataset) yields (main_ddpir.py:117, 262-267).
    sweep   : run the reference's own lambda/zeta sweep statements (main_ddpir.py:549-581) instead of one direct call.
    ns_over : {name: object} bound in test_rho's globals INSTEAD of the reference's own (`utils_model`, `sr`, `Resizer` ...): recording wrappers
              around the real plugs (oracle/gen_golden_shim_trace.py)
    returns : (list of x_0 tensors in the order test_rho produced them, config, test_results_ave)
    """
    md = main_module()
    pieces = main_pieces()
    captured = []

    def tensor2uint_batch(img):
        captured.append(img.detach().clone())
        return md.util.tensor2uint_batch(img)

    util = _Proxy(md.util, imsave_batch=lambda *a, **k: None, tensor2uint_batch=tensor2uint_batch)
    um = md.utils_model
    if trace is not None:
        def model_fn(x, *a, **k):
            out = md.utils_model.model_fn(x, *a, **k)
            trace.append(("model_fn_in", x.detach().clone()))
            trace.append(("model_fn_out", tuple(o.detach().clone() for o in out) if isinstance(out, tuple) else out.detach().clone()))
            return out

        def grad_and_value(**k):
            g, v = md.utils_model.grad_and_value(**k)
            trace.append(("norm_grad", g.detach().clone()))
            return g, v
        um = _Proxy(md.utils_model, model_fn=model_fn, grad_and_value=grad_and_value)

    with tempfile.TemporaryDirectory() as cwd:
        config = reference_config(ydict, cwd)
        for k_, v_ in (direct or {}).items():
            setattr(config, k_, v_)
        device = torch.device("cpu")
        config.device = device
        logger = logging.getLogger("reference_test_rho")
        logger.addHandler(logging.NullHandler())
        logger.propagate = False
        ns = dict(vars(md))
        ns.update(config=config, device=device, logger=logger, dataloader=batches, model=model, diffusion=diffusion,
                  util=util, utils_model=um,
                  test_results_ave=OrderedDict(psnr_sf=[], psnr_y_sf=[]))
        ns.update(ns_over or {})
        _exec(pieces["schedule"], ns)           # betas ... reduced_alpha_cumprod, config.noise_model_t, config.t_start
        _exec(pieces["freeze"], ns)             # requires_grad = False unless DPS_y0
        _exec(pieces["test_rho_deblur" if config.task == "deblur" else "test_rho"], ns)           # def test_rho(config)
        with patched_randn_like(noise_fn):
            if sweep:
                _exec(pieces["sweep"], ns)
            else:
                ns["test_results_ave"] = ns["test_rho"](config)
    return captured, config, ns["test_results_ave"]


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def restore_ref(model, diffusion, cfg, y, k=None, mask=None, noise_fn=None, y_label=None, trace=None, gt_u8=None, ns_over=None):
    """One batch through the reference's test_rho; `cfg` is an oracle.diffpir_oracle.LoopConfig (the YAML keys that reach the loop,
    already in test_rho's units: lambda_/zeta are the values AFTER the sweep's multipliers, noise_level_img is /255).
    y: [B,3,h,w] in [0,1]; k: [B,1,kh,kw]; mask: [B,3,H,W].  Returns x_0 [B,3,H,W] in [0,1] (main_ddpir.py:470)."""
    over = dict(iter_num=int(cfg.iter_num), noise_level_img=float(cfg.noise_level_img) * 255.0, lambda_=float(cfg.lambda_), zeta=float(cfg.zeta),
                eta=float(cfg.eta), guidance_scale=float(cfg.guidance_scale), sf=int(cfg.sf), skip_type=cfg.skip_type,
                num_train_timesteps=int(cfg.T), beta_start=float(cfg.beta_start), beta_end=float(cfg.beta_end),
                generate_mode=cfg.generate_mode, sub_1_analytic=bool(cfg.sub_1_analytic),
                noise_init_img=cfg.noise_init_img if cfg.noise_init_img == "max" else float(cfg.noise_init_img),
                ddim_sample=bool(cfg.ddim_sample), batch_size=int(y.shape[0]))
    if cfg.task == "sr":
        over.update(sr_mode=cfg.sr_mode, inIter=int(cfg.inIter), gamma=float(cfg.gamma))
    yd = yaml_for(cfg.task, **over)
    B = y.shape[0]
    H, W = y.shape[2] * cfg.sf, y.shape[3] * cfg.sf
    img_H = torch.zeros((B, H, W, 3), dtype=torch.uint8) if gt_u8 is None else torch.as_tensor(gt_u8)
    img_L = _nhwc(y.detach().float())
    names = [f"{i}.png" for i in range(B)]
    kk = torch.ones((B, 1, 1, 1, 1)) if k is None else k.detach().float()[:, 0]          # main_ddpir.py:74: dummy kernel otherwise
    mm = torch.ones_like(img_L) if mask is None else _nhwc(mask.detach().float())
    net = model if y_label is None else _Labelled(model, y_label)
    # parse_args_and_config re-derives sigma = max(0.001, noise_level_img/255) from the YAML; x255 then /255 may move the last bit of
    # noise_level_img, so the exact value of the case is put back (it is what the fixtures of rounds 1-4 were generated with)
    direct = dict(noise_level_img=float(cfg.noise_level_img), noise_level_model=float(cfg.noise_level_img), sigma=max(0.001, float(cfg.noise_level_img)))
    outs, config, _ = run_test_rho(yd, [(img_H, img_L, names, kk, mm)], net, diffusion, noise_fn=noise_fn, trace=trace, direct=direct, ns_over=ns_over)
    assert len(outs) == 1
    return outs[0]


# ------------------------------------------------------------------------------------------ the whole main()
def run_main(ydict, state_dict, images, noise_fn=None, build_model=None):
    """The reference's main() end to end in a scratch cwd: kernels/ -> the reference's kernel files, testsets/<testset_name>/ -> `images`
    ({file name: path or u8 HxWx3 array}), model_zoo/<model_name>.pt -> `state_dict`.  Everything runs: parse_args_and_config,
    CustomDataset.__getitem__ (cv2 image read, the shipped .mat kernels through hdf5storage.loadmat, degradation, np.random noise under the
    config seed), DataLoader collation, create_model_and_diffusion + torch.load + load_state_dict, test_rho, the lambda/zeta sweep.
    `main` is compiled from the source file's own FunctionDef; for task 'deblur' the one crashing statement is removed (_DropQ1, the
    same single edit as above).  `build_model(**kw)` replaces create_model_and_diffusion when the topology is not one of the two
    main() knows (tiny test networks).  Returns dict(x0=[x_0 per test_rho batch ...], batches=[(img_H, img_L, names, k, mask) ...])."""
    from PIL import Image
    md = main_module()
    captured, batches = [], []

    def tensor2uint_batch(img):
        captured.append(img.detach().clone())
        return md.util.tensor2uint_batch(img)

    real_loader = md.DataLoader

    class RecordingLoader:
        def __init__(self, dataset, **kw):
            self.items = list(real_loader(dataset, **kw))         # the dataset's np.random draws happen here, once, in index order

        def __iter__(self):
            for b in self.items:
                batches.append(b)
                yield b

    with open(MAIN_PY) as f:
        tree = ast.parse(f.read(), filename=MAIN_PY)
    main_fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    if ydict["task"] == "deblur":
        main_fn = _DropQ1().apply(main_fn)
    ns = dict(vars(md))
    ns.update(util=_Proxy(md.util, imsave_batch=lambda *a, **k: None, tensor2uint_batch=tensor2uint_batch), DataLoader=RecordingLoader)
    if build_model is not None:
        ns["create_model_and_diffusion"] = build_model
    _exec([main_fn], ns)
    # parse_args_and_config is a module-level function of md: it resolves `util` / `torch` in md's globals (the real ones) -- fine
    with tempfile.TemporaryDirectory() as cwd:
        os.symlink(os.path.join(ref_import.REF_ROOT, "kernels"), os.path.join(cwd, "kernels"))
        tdir = os.path.join(cwd, "testsets", ydict["testset_name"])
        os.makedirs(tdir)
        for name, src in images.items():
            if isinstance(src, str):
                os.symlink(src, os.path.join(tdir, name))
            else:
                Image.fromarray(src).save(os.path.join(tdir, name))
        os.makedirs(os.path.join(cwd, "model_zoo"))
        torch.save(state_dict, os.path.join(cwd, "model_zoo", ydict["model_name"] + ".pt"))
        path = os.path.join(cwd, "opt.yaml")
        with open(path, "w") as f:
            yaml.safe_dump(dict(ydict, cwd=cwd), f)
        argv = sys.argv
        sys.argv = ["main_ddpir.py", "--opt", path]
        try:
            with patched_randn_like(noise_fn):
                ns["main"]()
        finally:
            sys.argv = argv
    return dict(x0=captured, batches=batches)


# ------------------------------------------------------------------------------------------ schedule tables from the reference's statements
def reference_tables(ydict, direct=None):
    """rhos / sigmas / sigma_ks (main_ddpir.py:274-286) and the driver tables (:184-200) for one YAML, by executing the reference's own
    statements: the schedule piece of main() and, out of test_rho's batch loop, `model_out_type = ...` plus the statements from
    `sigmas = []` to the `rhos, sigmas, sigma_ks = torch.tensor(...)` line.  Returns numpy arrays + t_start."""
    md = main_module()
    pieces = main_pieces()
    fn = pieces["test_rho"][0]
    loop = next(n for n in fn.body if isinstance(n, ast.For) and _mentions(n.iter, "dataloader"))
    i_mot = next(i for i, n in enumerate(loop.body) if "model_out_type" in _names_assigned(n))
    i_sig = next(i for i, n in enumerate(loop.body) if "sigmas" in _names_assigned(n))
    i_end = next(i for i, n in enumerate(loop.body) if {"rhos", "sigmas", "sigma_ks"} <= _names_assigned(n))
    with tempfile.TemporaryDirectory() as cwd:
        config = reference_config(ydict, cwd)
        for k_, v_ in (direct or {}).items():
            setattr(config, k_, v_)
        config.device = torch.device("cpu")
        ns = dict(vars(md))
        ns.update(config=config, device=config.device)
        _exec(pieces["schedule"], ns)
        _exec([loop.body[i_mot]] + loop.body[i_sig:i_end + 1], ns)
    return dict(rhos=ns["rhos"].numpy(), sigmas=ns["sigmas"].numpy(), sigma_ks=ns["sigma_ks"].numpy(),
                reduced=ns["reduced_alpha_cumprod"].numpy(), sqrt_ac=ns["sqrt_alphas_cumprod"].numpy(),
                sqrt_1m_ac=ns["sqrt_1m_alphas_cumprod"].numpy(), t_start=int(config.t_start),
                lines=(loop.body[i_sig].lineno, loop.body[i_end].end_lineno))


def reference_step_trace(ydict, shape, k=None, direct=None, sweep=False):
    """(t_i, tau) of every step test_rho takes for a YAML (task deblur / sr-blur), with NO network: utils_model.model_fn is replaced by a
    recorder that returns zeros (its `noise_level` argument is mapped to t the way model_fn itself does, utils_model.py:215-217) and
    sr.data_solution by a recorder of its `alpha` argument (= tau, main_ddpir.py:389, 397)."""
    md = main_module()
    pieces = main_pieces()
    ts, taus = [], []

    def model_fn(x, noise_level, alphas_cumprod=None, **kw):
        red = torch.div(torch.sqrt(1. - alphas_cumprod), torch.sqrt(alphas_cumprod))
        ts.append(int(md.utils_model.find_nearest(red, noise_level / 255.)))
        return torch.zeros_like(x)

    def data_solution(x, FB, FBC, F2B, FBFy, alpha, sf):
        taus.append(float(alpha.reshape(-1)[0]))
        return x
    B, _, h, w = shape
    sf = int(ydict.get("sf", 1))
    batch = (torch.zeros((B, h * sf, w * sf, 3), dtype=torch.uint8), torch.zeros((B, h, w, 3)), [f"{i}.png" for i in range(B)],
             torch.ones((B, 3, 3)) / 9 if k is None else k, torch.ones((B, h, w, 3)))
    with tempfile.TemporaryDirectory() as cwd:
        config = reference_config(ydict, cwd)
        for k_, v_ in (direct or {}).items():
            setattr(config, k_, v_)
        config.device = torch.device("cpu")
        logger = logging.getLogger("reference_test_rho")
        logger.addHandler(logging.NullHandler()); logger.propagate = False
        ns = dict(vars(md))
        ns.update(config=config, device=config.device, logger=logger, dataloader=[batch], model=None, diffusion=None,
                  util=_Proxy(md.util, imsave_batch=lambda *a, **k: None), utils_model=_Proxy(md.utils_model, model_fn=model_fn),
                  sr=_Proxy(md.sr, data_solution=data_solution), test_results_ave=OrderedDict(psnr_sf=[], psnr_y_sf=[]))
        _exec(pieces["schedule"], ns)
        _exec(pieces["test_rho_deblur" if config.task == "deblur" else "test_rho"], ns)
        if sweep:                                   # main_ddpir.py:548-580: sr lambda x {2..12}, deblur lambda x 7 / zeta x 3, inpaint x 1
            _exec(pieces["sweep"], ns)
        else:
            ns["test_rho"](config)
    return np.array(ts, np.int64), np.array(taus, np.float32)
