timeout -k 10 300 python3 -m pytest tests/test_render.py -q -m gpu -x -k "common_planes" 2>&1 | tail -3
