"""Schedulers used by the fitting harnesses (mirror of the reference package layout)."""
