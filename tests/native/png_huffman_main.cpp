// Length-limited Huffman construction of the PNG writer (csrc/png_encode.cpp: huff_lengths) under adversarial
// frequency sets (powers of two, Fibonacci weights, one heavy symbol, few symbols): every code must be COMPLETE (inflate
// rejects under- and over-subscribed literal/length and code-length codes), within its length limit, and monotone.
// The .cpp is included directly: the function is internal to the library.
#include "png_encode.cpp"
#include <cstdio>
#include <random>
using namespace mic;
int main(){
  std::mt19937_64 g(7);
  long bad=0, over=0;
  for(int trial=0; trial<200000; ++trial){
    int n = (trial%3==0)?19:((trial%3==1)?30:286); int max_len = n==19?7:15;
    uint32_t freq[286]={0};
    int kind = trial % 5;
    int used = 1 + g()% n;
    for(int i=0;i<used;++i){
      int s = g()%n;
      uint32_t f;
      switch(kind){
        case 0: f = 1 + g()%1000; break;
        case 1: f = 1u << (g()%24); break;                // powers of two: deep trees
        case 2: { static uint64_t fib[40]; if(!fib[1]){fib[0]=1;fib[1]=1;for(int k=2;k<40;++k)fib[k]=fib[k-1]+fib[k-2];} f=(uint32_t)std::min<uint64_t>(fib[g()%38], 4000000000ull); break; }
        case 3: f = (g()%10==0)? 1000000 : 1; break;
        default: f = 1; break;
      }
      freq[s]=f;
    }
    uint8_t lens[286];
    huff_lengths(freq,n,max_len,lens, n==19);
    int m=0; long long kraft=0; bool lim=true;
    for(int i=0;i<n;++i){ if(freq[i]){++m; if(!lens[i]) {lim=false;} } if(lens[i]){ if(lens[i]>max_len) lim=false; kraft += 1LL<<(max_len-lens[i]); } }
    bool ok = lim && (m==1 ? (n==19 ? kraft==(1LL<<max_len) : kraft==(1LL<<(max_len-1))) : kraft==(1LL<<max_len));
    if(!ok){ if(bad<5) printf("BAD trial %d n %d m %d kraft %lld\n",trial,n,m,kraft); ++bad; }
    // optimality sanity: more frequent symbols never get longer codes
    for(int i=0;i<n && ok;++i) for(int j=0;j<n;++j) if(freq[i]>freq[j] && freq[j] && lens[i]>lens[j]) { ++over; i=n; break; }
  }
  printf("bad=%ld monotonic_violations=%ld\n",bad,over);
}
