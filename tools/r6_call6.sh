#!/bin/bash
# Round-6 GPU call 6: the measurement pass on the final code.
bash tools/prof_r6.sh all
