#!/usr/bin/env python3
"""Same entry point under the reference's file name (gen_3dphoto_dynamic_v2.py, the only generator the reference ships and the
one scripts/gen_train_kitti15_v2.sh calls): identical flags, defaults and output layout - see gen_3dphoto_dynamic.py."""
import sys

from gen_3dphoto_dynamic import main

if __name__ == "__main__":
    sys.exit(main())
