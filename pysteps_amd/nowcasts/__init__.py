"""Callers of the advection operators that are thin enough to mirror."""
