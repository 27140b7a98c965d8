#!/usr/bin/env python3
"""build-index.py DIR/ [DIR/ ...] — drop-in for the reference's indexer on MI355X (see cli-p_amd/indexer.py)."""
import sys

import clipmi
from clipmi.indexer import candidates, encode_directories, finalise, main  # noqa: F401

if __name__ == "__main__":
    main(sys.argv[1:])
