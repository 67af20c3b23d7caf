timeout -k 5 900 python -m pytest tests -m gpu -x -q -s -k "stress or mixed_cap" 2>&1 | grep -v "^$" | tail -8
timeout -k 5 600 python -m pytest tests -m gpu -x -q -k "not stress and not mixed_cap" 2>&1 | tail -3
