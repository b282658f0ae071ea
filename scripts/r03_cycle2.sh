#!/bin/bash
python scripts/time_upload.py 2>&1 | grep -v amdgpu
