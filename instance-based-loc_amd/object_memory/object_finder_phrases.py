"""Floor test of the memory build (/root/reference/object_memory/object_finder_phrases.py:19-36): an object whose NAME LIST
contains one of these words verbatim goes to the floor slot."""

FLOOR_WORDS = ("floor", "ground", "earth")


def check_if_floor(texts):
    return any(word in texts for word in FLOOR_WORDS)
