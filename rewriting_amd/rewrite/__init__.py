"""Host-side mirror of the reference's ``rewrite`` package (rewrite/ganrewrite.py)."""
