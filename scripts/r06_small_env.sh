# Round 6: small meshes after the segment floor went from eight elements to two
python3 scripts/r06_small.py
