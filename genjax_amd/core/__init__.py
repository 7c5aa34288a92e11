from .choice_map import (ChoiceMap, ChoiceMapBuilder, ChoiceMapNoValueAtAddress, Selection,
                         SelectionBuilder)
from .mask import Mask

__all__ = ["ChoiceMap", "ChoiceMapBuilder", "ChoiceMapNoValueAtAddress", "Selection",
           "SelectionBuilder", "Mask"]
