"""Decodes the reference's earthmap.jpg (used by examples/final_scene.rs:147 through
Texture::image, src/textures.rs:41-44) into a binary PPM.  Run once where the reference
checkout and Pillow are available; the PPM is committed because neither travels to the GPU box
and the C++ side carries no JPEG decoder.  The file is 4:4:4 baseline JPEG, so decoders agree
to within the IDCT rounding (about 1 LSB)."""
import sys
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/earthmap.jpg"
dst = sys.argv[2] if len(sys.argv) > 2 else "raytracing_amd/assets/earthmap.ppm"
im = Image.open(src).convert("RGB")
with open(dst, "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % im.size)
    f.write(im.tobytes())
print(dst, im.size)
