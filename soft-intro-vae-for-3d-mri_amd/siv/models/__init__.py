"""Model definitions with the reference's module paths (models.models, models.vaemodel,
models.lossf) and ``state_dict`` layout."""
