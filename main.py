#!/usr/bin/env python3
"""Entry point kept from the reference: `python main.py --mode=test --root_dataset ... --model_dir ...`."""
from rcvpose_amd.main import main

if __name__ == "__main__":
    main()
