#!/usr/bin/env python3
"""Golden image fixtures: the reference authors' own renders shipped next to the scene files
(/root/reference/scenes/torus/{lmc,h2mc}_timeuse_*.exr, 1024x768 RGB half ZIP), decoded with the product's EXR reader and
box-downsampled 4x to 256x192 (keeps the fixture small and averages the MCMC noise of the 245-spp originals).
Run in the build container (needs /root/reference and a built liblmc_hip.so; no GPU: the image codec is host code)."""
import importlib, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
REF = "/root/reference/scenes/torus"
out = {}
for key, fn in (("lmc", "lmc_timeuse_44.689152s.exr"), ("h2mc", "h2mc_timeuse_45.381592s.exr")):
    img = p.read_image(os.path.join(REF, fn))
    h, w, _ = img.shape
    out[key] = img.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3)).astype(np.float32)
    print(key, img.shape, float(img.mean()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"), **out)
# veach-door: /root/reference/scenes/veachdoor/{lmc,h2mc}_timeuse_*.exr, 1280x720 -> 320x180
REF = "/root/reference/scenes/veachdoor"
out = {}
for key, fn in (("lmc", "lmc_timeuse_30.236183s.exr"), ("h2mc", "h2mc_timeuse_32.686382s.exr")):
    img = p.read_image(os.path.join(REF, fn))
    h, w, _ = img.shape
    out[key] = img.reshape(h // 4, 4, w // 4, 4, 3).mean(axis=(1, 3)).astype(np.float32)
    print("veachdoor", key, img.shape, float(img.mean()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "veachdoor_ref_images_320x180.npz"), **out)
