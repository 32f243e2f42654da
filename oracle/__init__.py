"""ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (mgnet_amd) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIBS = {}


def build(force=False):
    """Compile the C oracle with gcc (a few seconds)."""
    need = force or not all(os.path.exists(os.path.join(_BUILD, f"liboracle_{s}.so")) for s in ("f32", "f64"))
    if not need:
        srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_oracle.c")]
        newest = max(os.path.getmtime(s) for s in srcs)
        need = any(os.path.getmtime(os.path.join(_BUILD, f"liboracle_{s}.so")) < newest for s in ("f32", "f64"))
    if need:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, stdout=subprocess.DEVNULL)


def _lib(prec):
    if prec not in _LIBS:
        path = os.path.join(_BUILD, f"liboracle_{prec}.so")
        if not os.path.exists(path):
            build()
        _LIBS[prec] = ctypes.CDLL(path)
    return _LIBS[prec]


def _dt(prec):
    return np.float32 if prec == "f32" else np.float64


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads(prec="f32"):
    f = getattr(_lib(prec), f"orc_num_threads_{prec}")
    f.restype = ctypes.c_int
    return f()


def pose_vec2mat(vec6, prec="f32"):
    """[B,6] -> (R [B,3,3], t [B,3])   (pose_utils.py:9-51, pose.py:40-46)"""
    dt = _dt(prec)
    vec6 = _c(vec6, dt).reshape(-1, 6)
    R = np.zeros((len(vec6), 3, 3), dt)
    t = np.zeros((len(vec6), 3), dt)
    f = getattr(_lib(prec), f"orc_pose_vec2mat_{prec}")
    for b in range(len(vec6)):
        f(_p(vec6[b]), _p(R[b]), _p(t[b]))
    return R, t


def kinv(K, prec="f32"):
    dt = _dt(prec)
    K = _c(K, dt).reshape(-1, 3, 3)
    out = np.zeros_like(K)
    f = getattr(_lib(prec), f"orc_kinv_{prec}")
    for b in range(len(K)):
        f(_p(K[b]), _p(out[b]))
    return out


def inv2depth(inv, prec="f32"):
    dt = _dt(prec)
    inv = _c(inv, dt)
    out = np.zeros_like(inv)
    getattr(_lib(prec), f"orc_inv2depth_{prec}")(_p(inv), _p(out), ctypes.c_long(inv.size))
    return out


def view_synthesis(ref, inv_depth, K, vec, prec="f32"):
    """ref [B,3,H,W], inv_depth [B,1,H,W], K [B,3,3], vec [B,6] -> warped [B,3,H,W]"""
    dt = _dt(prec)
    ref, inv_depth, K, vec = _c(ref, dt), _c(inv_depth, dt), _c(K, dt), _c(vec, dt)
    B, _, H, W = ref.shape
    out = np.zeros_like(ref)
    getattr(_lib(prec), f"orc_view_synthesis_{prec}")(_p(ref), _p(inv_depth), _p(K), _p(vec), B, H, W, _p(out))
    return out


def ssim(x, y, prec="f32"):
    dt = _dt(prec)
    x, y = _c(x, dt), _c(y, dt)
    H, W = x.shape[-2:]
    out = np.zeros_like(x)
    getattr(_lib(prec), f"orc_ssim_{prec}")(_p(x), _p(y), int(x.size // (H * W)), H, W, _p(out))
    return out


def photometric(est, img, ssim_w=0.85, prec="f32"):
    dt = _dt(prec)
    est, img = _c(est, dt), _c(img, dt)
    B, _, H, W = est.shape
    out = np.zeros((B, 1, H, W), dt)
    f = getattr(_lib(prec), f"orc_photometric_{prec}")
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                  ctypes.c_float if prec == "f32" else ctypes.c_double, ctypes.c_void_p]
    f(_p(est), _p(img), B, H, W, ssim_w, _p(out))
    return out


def calc_smoothness(inv, img, prec="f32"):
    dt = _dt(prec)
    inv, img = _c(inv, dt), _c(img, dt)
    B, _, H, W = img.shape
    sx = np.zeros((B, 1, H, W - 1), dt)
    sy = np.zeros((B, 1, H - 1, W), dt)
    getattr(_lib(prec), f"orc_calc_smoothness_{prec}")(_p(inv), _p(img), B, H, W, _p(sx), _p(sy))
    return sx, sy


def reproj_loss(inv, img, prev, nxt, mask, K, poses, ssim_w=0.85, photo_w=1.0, smooth_w=0.001,
                want_grad=True, g_photo=1.0, g_smooth=1.0, want_minmap=False, prec="f32", automask=True, reduce_op="min",
                padding_mode="zeros"):
    """MultiViewPhotometricLoss.forward (+backward) -- loss.py:111-154.

    inv: list of [B,1,H,W]; img/prev/nxt [B,3,H,W]; mask [B,1,H,W] bool or None; K [B,3,3] (or
    [B,4,4] camera_matrix, of which the top-left 3x3 is taken like loss.py:122); poses [B,2,6].
    Returns dict(loss_photometric, loss_smoothness, d_inv=[...], d_poses, minmap=[...]).
    Gradients are those of g_photo*loss_photometric + g_smooth*loss_smoothness.
    """
    dt = _dt(prec)
    real = ctypes.c_float if prec == "f32" else ctypes.c_double
    n = len(inv)
    inv = [_c(a, dt) for a in inv]
    img, prev, nxt, poses = _c(img, dt), _c(prev, dt), _c(nxt, dt), _c(poses, dt)
    K = np.asarray(K)
    if K.shape[-1] == 4:
        K = K[:, :3, :3]
    K = _c(K, dt)
    B, _, H, W = img.shape
    m8 = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
    losses = np.zeros(2, dt)
    d_inv = [np.zeros_like(a) for a in inv]
    d_poses = np.zeros((B, 2, 6), dt)
    minmap = [np.zeros((B, 1, H, W), dt) for _ in range(n)] if want_minmap else None
    PP = ctypes.c_void_p * n
    inv_pp = PP(*[a.ctypes.data for a in inv])
    dinv_pp = PP(*[a.ctypes.data for a in d_inv])
    mm_pp = PP(*[a.ctypes.data for a in minmap]) if want_minmap else None
    fo = getattr(_lib(prec), f"orc_reproj_options_{prec}")
    fo.restype, fo.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    if fo(int(bool(automask)), {"min": 0, "mean": 1}[reduce_op], {"zeros": 0, "border": 1, "reflection": 2}[padding_mode]) != 0:
        raise ValueError("automask_loss goes with photometric_reduce_op 'min' only (loss.py:105-109)")
    f = getattr(_lib(prec), f"orc_reproj_loss_{prec}")
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [real] * 3 + \
                 [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, real, real, ctypes.c_void_p, ctypes.c_void_p]
    rc = f(inv_pp, n, _p(img), _p(prev), _p(nxt), None if m8 is None else _p(m8), _p(K), _p(poses), B, H, W,
           ssim_w, photo_w, smooth_w, _p(losses), mm_pp, int(want_grad), g_photo, g_smooth, dinv_pp, _p(d_poses))
    if rc == -2:   # ssim_loss_weight == 0: "min" needs a reprojection mask, "mean" must not have one (the reference's own IndexError)
        raise IndexError("ssim_loss_weight=0: the 3-channel L1 maps cannot be indexed by this mask (loss.py:236-246)")
    if rc != 0:
        raise ValueError(f"orc_reproj_loss: bad arguments (rc={rc})")
    return {"loss_photometric": losses[0], "loss_smoothness": losses[1], "d_inv": d_inv, "d_poses": d_poses,
            "minmap": minmap}
