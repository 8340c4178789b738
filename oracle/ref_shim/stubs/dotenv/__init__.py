"""Import-only stub so /root/reference's surya.settings imports in this container (no python-dotenv)."""
def find_dotenv(*a, **k):
    return ""
def load_dotenv(*a, **k):
    return False
