"""Trainers with the reference's module path (utils.my_trainer)."""
