"""Drop-in `animatediff` package: the reference's import paths, backed by the MI355X engine."""
