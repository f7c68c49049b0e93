python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4
