from .seq2seq import Seq2Seq
from .seq2seq_embeddings import Seq2SeqEmbeddings
