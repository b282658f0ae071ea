#!/bin/bash
python scripts/time_run_layouts.py 2>&1 | grep -v amdgpu.ids
