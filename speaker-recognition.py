#!/usr/bin/env python3
"""Entry point with the reference's name (src/speaker-recognition.py); see
speaker-recognition_amd/cli.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from speaker_recognition_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main()
