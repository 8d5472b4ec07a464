      seqfile = ../data/mtCDNApri.aa
     treefile = ../data/mtCDNApri.trees
      seqtype = 2
        model = 3
   aaRatefile = ../data/jones.dat
    fix_alpha = 1
        alpha = 0
    cleandata = 1
