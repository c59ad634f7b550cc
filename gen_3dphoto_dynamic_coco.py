#!/usr/bin/env python3
"""The entry point scripts/gen_coco.sh calls (the file itself is not in the reference tree; its library half is
utils/utils_coco.py): gen_3dphoto_dynamic.py with the COCO pose sampler (utils/utils_coco.py:121-156) as the default."""
import sys

from gen_3dphoto_dynamic import main

if __name__ == "__main__":
    argv = sys.argv[1:]
    if not any(a == "--poses" or a.startswith("--poses=") for a in argv):
        argv = ["--poses", "coco"] + argv
    sys.exit(main(argv))
