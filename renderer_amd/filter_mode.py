"""Sampler filter (reference src/filter_mode.rs:2-21)."""
import enum


class FilterMode(enum.IntEnum):
    Linear = 0
    Nearest = 1

    def is_linear(self):
        return self is FilterMode.Linear
