"""File-list harness: the Python twin of the reference's offline feeder `StereonetNode::RunImglistFeedInfer`
(stereonet_infer/src/stereonet_node.cpp:820-976) as a plain command line, plus the stereo metrics (EPE, bad-N,
D1) needed to score the output against dataset ground truth.

    python -m hobot_stereonet_amd.filelist --model m.snw --left left.list --right right.list \
        [--gt gt.list] [--out out_dir] [--precision auto|f16|f16x3|fp32]

Per frame i (same order as the reference): read left[i] / right[i] (8-bit colour image) -> BGR -> NV12
(`images.bgr_to_nv12`) -> side-by-side frame -> `sn_infer_sbs_nv12` (split + pre-processing + network on the
GPU) -> int32 wire tensor + float disparity.  Error behaviour mirrors the reference: an unreadable list, a
missing image or lists of different length stop the run before any inference.  There is no CPU path.
"""
import argparse
import json
import os
import sys
from typing import Dict, List, Optional

import numpy as np

from . import images


class FileListError(RuntimeError):
    pass


def read_list(path: str) -> List[str]:
    """One image path per line; every entry must exist (stereonet_node.cpp:832-878)."""
    if not os.path.isfile(path):
        raise FileListError(f"Open file failed: {path}")
    out = []
    with open(path) as f:
        for line in f.read().splitlines():
            name = line.rstrip("\r ")
            if not os.path.exists(name):
                raise FileListError(f"File is not exist! img_name: {name}")
            out.append(name)
    return out


def read_pair_lists(left_list: str, right_list: str):
    left, right = read_list(left_list), read_list(right_list)
    if len(left) != len(right):
        raise FileListError(f"Imgs size error! left_imgs.size: {len(left)}, right_imgs.size: {len(right)}")
    return left, right


# ---- metrics ----------------------------------------------------------------------------------------------
def epe(pred: np.ndarray, gt: np.ndarray, valid: Optional[np.ndarray] = None) -> float:
    """Mean absolute disparity error over the valid pixels (end-point error of a 1-D flow)."""
    err = np.abs(pred.astype(np.float64) - gt.astype(np.float64))
    if valid is not None:
        err = err[valid]
    return float(err.mean()) if err.size else float("nan")


def bad_px(pred: np.ndarray, gt: np.ndarray, thresh: float, valid: Optional[np.ndarray] = None) -> float:
    """Fraction of valid pixels whose error exceeds `thresh` px (SceneFlow/Middlebury bad-N)."""
    err = np.abs(pred.astype(np.float64) - gt.astype(np.float64))
    if valid is not None:
        err = err[valid]
    return float((err > thresh).mean()) if err.size else float("nan")


def d1(pred: np.ndarray, gt: np.ndarray, valid: Optional[np.ndarray] = None) -> float:
    """KITTI 2015 D1: error > 3 px AND > 5 % of the true disparity."""
    p, g = pred.astype(np.float64), gt.astype(np.float64)
    err = np.abs(p - g)
    bad = (err > 3.0) & (err > 0.05 * np.abs(g))
    if valid is not None:
        bad = bad[valid]
    return float(bad.mean()) if bad.size else float("nan")


def score(pred: np.ndarray, gt: np.ndarray, valid: Optional[np.ndarray], dmax: Optional[float] = None) -> Dict[str, float]:
    if dmax is not None:      # the usual protocol: pixels beyond the search range are not scored
        valid = (gt < dmax) if valid is None else (valid & (gt < dmax))
    return {"epe": epe(pred, gt, valid), "bad1": bad_px(pred, gt, 1.0, valid), "bad3": bad_px(pred, gt, 3.0, valid),
            "d1": d1(pred, gt, valid), "valid_px": int(valid.sum()) if valid is not None else int(gt.size)}


# ---- the feeder ---------------------------------------------------------------------------------------------
def run_imglist(engine, left_list: str, right_list: str, out_dir: Optional[str] = None,
                gt_list: Optional[str] = None, log=None) -> List[dict]:
    """Feeds every (left[i], right[i]) pair through `engine` (api.StereoNetHIP).  Returns one record per frame:
    {"frame_id", "left", "right", "raw" (int32 HxW), "disp" (float32 HxW)[, "metrics"]}; with `out_dir` also
    writes <i>.raw.bin, <i>.disp.pfm and <i>.depth.ppm (the render node's colour map)."""
    left, right = read_pair_lists(left_list, right_list)
    gts = read_list(gt_list) if gt_list else None
    if gts is not None and len(gts) != len(left):
        raise FileListError(f"Imgs size error! left_imgs.size: {len(left)}, gt.size: {len(gts)}")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    w, h = engine.width, engine.height
    results = []
    for i, (lp, rp) in enumerate(zip(left, right)):
        if log:
            log(f"Feed {i}/{len(left)}")
        eyes = []
        for p in (lp, rp):
            bgr = images.imread_bgr(p)
            if bgr.shape[:2] != (h, w):
                raise FileListError(f"BGRToNv12 Fail: {p} is {bgr.shape[1]}x{bgr.shape[0]}, model input is {w}x{h}")
            eyes.append(images.bgr_to_nv12(bgr))
        disp, raw = engine.infer_sbs_nv12(images.sbs_from_eyes(eyes[0], eyes[1], w, h))
        rec = {"frame_id": str(i), "left": lp, "right": rp, "raw": raw, "disp": disp}
        if gts is not None:
            gt, valid = images.read_disparity(gts[i])
            rec["metrics"] = score(disp, gt, valid, float(engine.dmax))
        if out_dir:
            raw.tofile(os.path.join(out_dir, f"{i}.raw.bin"))
            images.write_pfm(os.path.join(out_dir, f"{i}.disp.pfm"), disp)
            from . import render
            _, depth = render.disparity_and_depth(raw.view(np.uint32))
            images.write_ppm(os.path.join(out_dir, f"{i}.depth.ppm"), render.colorize_depth(depth)[..., ::-1])
        results.append(rec)
    return results


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--model", required=True, help=".snw weight file (the role of hobot_stereonet.hbm)")
    ap.add_argument("--left", required=True)
    ap.add_argument("--right", required=True)
    ap.add_argument("--gt", default=None, help="optional list of ground-truth disparities (.pfm or 16-bit .png)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--precision", choices=["auto", "f16", "f16x3", "fp32"], default="auto")
    ap.add_argument("--device", type=int, default=-1)
    args = ap.parse_args(argv)
    from . import api
    prec = {"auto": api.PREC_AUTO, "f16": api.PREC_F16, "f16x3": api.PREC_F16X3, "fp32": api.PREC_FP32}[args.precision]
    try:
        read_pair_lists(args.left, args.right)          # fail on the lists before touching the GPU
        with api.StereoNetHIP(args.model, device=args.device, precision=prec) as eng:
            recs = run_imglist(eng, args.left, args.right, args.out, args.gt, log=lambda s: print(s, file=sys.stderr))
    except (FileListError, ValueError) as e:
        print(f"error: {e}", file=sys.stderr)
        return 5
    summary = {"frames": len(recs)}
    if args.gt and recs:
        for k in ("epe", "bad1", "bad3", "d1"):
            summary[k] = float(np.nanmean([r["metrics"][k] for r in recs]))
    print(json.dumps(summary))
    return 0


if __name__ == "__main__":
    sys.exit(main())
