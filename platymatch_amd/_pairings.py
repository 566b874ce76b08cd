"""The widget's eight hypotheses (_dock_widget.py:547-611) and how they pair up: the eight cost matrices hold four distinct sets
of terms (U11/U22, U12/U21, U13/U24, U14/U23 differ only in summation order), so pairing t = (hypothesis summed in natural order,
its twin).  Host constants only — importable without the native library."""
HYPOTHESES = ("11", "12", "13", "14", "21", "22", "23", "24")
PAIRINGS = ((0, 5), (1, 4), (2, 7), (3, 6))
TWINS = {twin: h for h, twin in PAIRINGS}            # twin -> the hypothesis whose duals it first tries
