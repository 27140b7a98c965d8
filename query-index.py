#!/usr/bin/env python3
"""query-index.py — drop-in for the reference's interactive search prompt on MI355X (see cli-p_amd/repl.py)."""
import clipmi
from clipmi.repl import HELP, Viewer, main, normalize, repl  # noqa: F401

if __name__ == "__main__":
    main()
