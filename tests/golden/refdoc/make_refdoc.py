#!/usr/bin/env python3
"""Copy the reference's own published pictures of ONE real run into tests/golden/refdoc/ (data, not source).

The reference's documentation shows, for one frame of a real sea sequence,

  stereo_input0.jpg            doc/src/static/img/  -- `stereo_input.jpg` as wass_stereo.cpp:833 writes it: the two padded
                                                       pictures handed to cv::StereoSGBM::compute, left on top of right
                                                       (render.hpp:152-163), full resolution (2837 x 2*1753, i.e. a
                                                       2197 x 1753 crop + MAX_DISPARITY = 640 columns of padding), JPEG
  disparity_stereo_output.png  doc/src/static/img/  -- the map that call returned after clean_and_convert_disparity, drawn by
                                                       render_disparity_float (render.hpp:101-136: (d - min) / (max - min) * 255)
                                                       and, in the version of the tool that made the picture, resized to
                                                       600 rows with INTER_LINEAR (the two commented lines :129-130)
  disparity_final_scaled.png   doc/src/static/img/  -- the same after the dilate / erode clean-up (wass_stereo.cpp:1017)

(documentation/stereo.html.md:58, documentation/getting_started.html.md:221-222).  They are the only outputs of the real
cv::StereoSGBM-based executable anywhere in the reference tree, so tests/test_refdoc_pin.py runs the oracle (and the GPU
path) on the first and compares with the other two.  The input is a lossy JPEG copy of what the reference read, and the
pictures are 8-bit renderings at a third of the resolution (one grey level = 2.5 px of disparity): this anchors image roles,
padding, crop, sign, scale, the clean-up and the SHAPE of the rejected regions on the real program's output; it cannot show
bit-exactness.

  stereo_input100.jpg, stereo_input-100.jpg (documentation/stereo.html.md:67-68) -- the same frame's `stereo_input.jpg` with
                                                       DISPARITY_OFFSET = 100 / -100: what row a1's padding rule
                                                       (wass_stereo.cpp:801-831) does to the two pictures.  Only a band of
                                                       32 rows of each half is kept (rows 800..831, stored losslessly as
                                                       stereo_input{+100,-100}_band.png: the left band on top of the right).

Run in the build container only:  python tests/golden/refdoc/make_refdoc.py
"""
import hashlib
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/doc/src/static/img"
for name in ("stereo_input0.jpg", "disparity_stereo_output.png", "disparity_final_scaled.png"):
    shutil.copyfile(os.path.join(SRC, name), os.path.join(HERE, name))
    os.chmod(os.path.join(HERE, name), 0o644)
    print(name, hashlib.sha256(open(os.path.join(HERE, name), "rb").read()).hexdigest())

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

for src, dst in (("stereo_input100.jpg", "stereo_input+100_band.png"), ("stereo_input-100.jpg", "stereo_input-100_band.png")):
    im = np.array(Image.open(os.path.join(SRC, src)))
    h = im.shape[0] // 2
    band = np.concatenate([im[800:832], im[h + 800:h + 832]], 0)
    Image.fromarray(band).save(os.path.join(HERE, dst), optimize=True)
    print(dst, band.shape, os.path.getsize(os.path.join(HERE, dst)))
