class OwnedIterator:  # only referenced in type annotations / isinstance checks of gpflow.models.training_mixins
    pass
