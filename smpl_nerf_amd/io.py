"""On-disk formats of the reference, so that datasets and checkpoints move between the two code bases
(SURVEY 8(f)-3).  Host side only (numpy / PIL / torch.save); no cv2 in this image, so PNGs are read with
PIL and flipped RGB -> BGR to reproduce cv2.imread's channel order.

  transforms.json   create_dataset.py:92-105, 129-134   {camera_angle_x, image_transform_map{name -> 4x4},
                                                         [image_pose_map{name -> [69]}, betas[10], expression[10]]}
  img_XXX.png       datasets/rays_from_images_dataset.py:39-43 (sorted glob, cv2.imread -> BGR uint8)
  model_*.pt        utils.py:267-289 (torch.save(state_dict) per model)
  scores            util/scores.py:11-48 (img2mse, img2psnr), utils.py:484-488 (mse2psnr)
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np


def read_image_bgr(path: str) -> np.ndarray:
    """cv2.imread(path) equivalent for 8-bit PNGs: uint8 [H,W,3] in BGR order (alpha dropped)."""
    from PIL import Image
    img = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return img[..., ::-1].copy()


def load_dataset(image_directory: str, transforms_file: str = None):
    """What RaysFromImagesDataset.__init__ / SmplNerfDataset.__init__ read
    (datasets/rays_from_images_dataset.py:32-47, datasets/smpl_nerf_dataset.py:37-60).
    Returns dict(images uint8 [F,H,W,3] BGR, poses fp64 [F,4,4], camera_angle_x, names, human_poses [F,69] or
    None, betas, expression)."""
    transforms_file = transforms_file or os.path.join(image_directory, "transforms.json")
    with open(transforms_file, "r") as f:
        td = json.load(f)
    tmap = td.get("image_transform_map")
    paths = sorted(glob.glob(os.path.join(image_directory, "*.png")))
    if not len(paths) == len(tmap):
        raise ValueError("Number of images in image_directory is not the same as number of transforms")
    names = [os.path.basename(p) for p in paths]
    images = np.stack([read_image_bgr(p) for p in paths])
    poses = np.stack([np.array(tmap[n], dtype=np.float64) for n in names])
    pmap = td.get("image_pose_map")
    human = np.stack([np.array(pmap[n], dtype=np.float32) for n in names]) if pmap else None
    return dict(images=images, poses=poses, camera_angle_x=td["camera_angle_x"], names=names, human_poses=human,
                betas=td.get("betas"), expression=td.get("expression"))


def write_dataset(directory: str, images_rgb_uint8, poses, camera_angle_x, human_poses=None, betas=None,
                  expression=None):
    """Writes img_XXX.png + transforms.json in the layout of create_dataset.save_split
    (create_dataset.py:86-134).  `images_rgb_uint8` [F,H,W,3] RGB like plt.imsave writes them."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    names = ["img_{:03d}.png".format(i) for i in range(len(images_rgb_uint8))]
    for n, im in zip(names, images_rgb_uint8):
        Image.fromarray(np.asarray(im, np.uint8), "RGB").save(os.path.join(directory, n))
    td = {"camera_angle_x": float(camera_angle_x),
          "image_transform_map": {n: np.asarray(p, np.float64).tolist() for n, p in zip(names, poses)}}
    if human_poses is not None:
        td["image_pose_map"] = {n: np.asarray(h, np.float32).tolist() for n, h in zip(names, human_poses)}
        td["betas"] = list(np.zeros(10).tolist() if betas is None else betas)
        td["expression"] = list(np.zeros(10).tolist() if expression is None else expression)
    with open(os.path.join(directory, "transforms.json"), "w") as fp:
        json.dump(td, fp)
    return names


def normalize_rgb(images_uint8) -> np.ndarray:
    """NormalizeRGB (datasets/transforms.py:33): /255 -> float32."""
    return (np.array(images_uint8) / 255.).astype(np.float32)


def save_run(save_dir: str, models, model_names):
    """utils.save_run (utils.py:282-283): one state_dict file per model, e.g. model_coarse.pt / model_fine.pt /
    model_warp_field.pt - loadable by the reference and by smpl_nerf_amd.nets alike."""
    import torch
    os.makedirs(save_dir, exist_ok=True)
    for model, name in zip(models, model_names):
        torch.save(model.state_dict(), os.path.join(save_dir, name))


def load_run(load_dir: str, models, model_names, map_location="cpu"):
    """Counterpart used by inference.py:123-131 / train.py:161-166."""
    import torch
    for model, name in zip(models, model_names):
        model.load_state_dict(torch.load(os.path.join(load_dir, name), map_location=map_location))


def img2mse(x, y) -> float:
    """util/scores.py:28: mean over all pixels and channels."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return float(np.mean((x - y) ** 2))


def img2psnr(x, y) -> float:
    """util/scores.py:47-48: -10 ln(mse) / ln(10)."""
    return float(-10. * np.log(img2mse(x, y)) / np.log(10.))


def mse2psnr(mse: float) -> float:
    """utils.py:484-488 (zero mse clamped to 1e-5)."""
    if mse == 0:
        mse = 1e-5
    return float(-10.0 * np.log10(mse))


def to_uint8_rgb(rgb_bgr_float) -> np.ndarray:
    """inference.py:261-263: clip to [0,1], *255, BGR -> RGB for saving."""
    img = np.clip(np.asarray(rgb_bgr_float), 0, 1) * 255
    return img.astype(np.uint8)[..., ::-1]
