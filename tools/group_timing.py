#!/usr/bin/env python3
"""Developer tool: smpc_group_optimize's own stage times (SMPC_GROUP_TIMING=1 prints every 64
batched ticks) for the multi-query configuration."""
import os, sys
os.environ["SMPC_GROUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_multi_query
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
print(time_multi_query(n, B, 64, 200, 256, 20))
