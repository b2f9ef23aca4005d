from .seq2seq import Seq2Seq
from .seq2seq_embeddings import Seq2SeqEmbeddings
from .seq2seq_residual import Seq2SeqResidualA, Seq2SeqResidualB, Seq2SeqResidualC
